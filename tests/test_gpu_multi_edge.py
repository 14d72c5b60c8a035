"""GPU: n-ary edges (BaseMultiEdge::constructQuadraticForm, /root/reference/g2o/core/base_multi_edge.hpp:170-222) through the
C ABI: one binary edge set per vertex PAIR of the edges, with the parts another pair already contributes switched off
(g2ohip_set_edge_set_parts).  Hpp, Hpl, Hll, b, chi2 and the solution are compared with the ORACLE's restatement of the n-ary
quadratic form (oracle/g2o_oracle.c, orc_add_multi_edge_set: computeUpperTriangleIndex block table, transposed helper blocks) --
itself pinned on the CPU against the dense NumPy assembly below (tests/test_oracle.py) -- and with that dense assembly (every
vertex's diagonal block and right-hand side once, every pair's off-diagonal block once, the edge's robust weight on all of
them: base_multi_edge.hpp:35-49 + robust_kernel_impl.cpp:65-78)."""
import numpy as np
import pytest

from openslam_g2o_amd import capi
from oracle import oracle as O

gpu = pytest.mark.gpu


def _huber(e2, delta):
    """rho(e2), rho'(e2) of RobustKernelHuber (robust_kernel_impl.cpp:65-78)."""
    if delta <= 0:
        return e2, 1.0
    dsqr = delta * delta
    if e2 <= dsqr:
        return e2, 1.0
    s = np.sqrt(e2)
    return 2 * s * delta - dsqr, delta / s


def _dense(n_tot, dims, offs, verts, J, omega, err, delta):
    """Dense H, b, chi2 of n-ary edges: verts [n][arity] (global vertex or -1), J[i] = [n][d][dim_i]."""
    H = np.zeros((n_tot, n_tot))
    b = np.zeros(n_tot)
    chi = 0.0
    n, ar = verts.shape
    for k in range(n):
        e = err[k]
        Om = omega[k]
        rho0, rho1 = _huber(float(e @ Om @ e), delta)
        chi += rho0
        for i in range(ar):
            vi = verts[k, i]
            if vi < 0:
                continue
            Ji = J[i][k]
            si = slice(offs[vi], offs[vi] + dims[vi])
            H[si, si] += rho1 * Ji.T @ Om @ Ji
            b[si] -= rho1 * Ji.T @ Om @ e
            for j in range(i + 1, ar):
                vj = verts[k, j]
                if vj < 0:
                    continue
                Jj = J[j][k]
                sj = slice(offs[vj], offs[vj] + dims[vj])
                blk = rho1 * Ji.T @ Om @ Jj
                H[si, sj] += blk
                H[sj, si] += blk.T
    return H, b, chi


def ternary_problem(with_landmark, seed=5):
    """400 three-vertex edges over 40 3-dof poses (+ 25 2-dof landmarks): vertex 0, 1 poses (some fixed: -1); vertex 2 a pose, or
    a landmark (EdgeSE2PointXYCalib-shaped: two poses + a point)."""
    rng = np.random.default_rng(seed)
    p, l, d = 3, 2, 2
    nP, nL, n = 40, (25 if with_landmark else 0), 400
    v = np.stack([rng.integers(-1, nP, n), rng.integers(0, nP, n), rng.integers(0, nL, n) + nP if with_landmark else rng.integers(0, nP, n)], 1)
    v[v[:, 0] == v[:, 1], 0] = -1                              # (distinct vertices per edge)
    if not with_landmark:
        v[(v[:, 2] == v[:, 1]) | (v[:, 2] == v[:, 0]), 2] = -1
    dims = np.array([p] * nP + [l] * nL)
    offs = np.concatenate([[0], np.cumsum(dims)[:-1]])
    J = [rng.normal(size=(n, d, p)), rng.normal(size=(n, d, p)), rng.normal(size=(n, d, l if with_landmark else p))]
    A = rng.normal(size=(n, d, d))
    omega = A @ A.transpose(0, 2, 1) + 0.5 * np.eye(d)
    err = rng.normal(size=(n, d)) * 1.5
    return dict(p=p, l=l, d=d, nP=nP, nL=nL, n=n, v=v.astype(np.int32), dims=dims, offs=offs, J=J, omega=omega, err=err, rng=rng)


def col(a):
    """[n][d][dim] -> [n][d * dim], column-major d x dim."""
    return np.ascontiguousarray(a.transpose(0, 2, 1)).reshape(len(a), -1)


def oracle_ternary(T, huber):
    """The same graph on the oracle: ONE n-ary edge set (BaseMultiEdge) + the unary priors that make every random topology
    positive definite."""
    p, l, nP, nL = T["p"], T["l"], T["nP"], T["nL"]
    o = O.OracleSolver(p, l, nP, nL, nL > 0)
    m = o.add_multi_edge_set(T["d"], T["v"])
    kp = o.add_edge_set(p, np.arange(nP))
    o.set_dims(kp, p, 0)
    kl = None
    if nL:
        kl = o.add_edge_set(l, np.arange(nL) + nP)
        o.set_dims(kl, l, 0)
    o.build_structure()
    o.set_multi_edge_data(m, [col(j) for j in T["J"]], col(T["omega"]), T["err"], huber, 1)
    Ip = np.tile(np.eye(p).reshape(-1), (nP, 1))
    o.set_edge_data(kp, Ip, None, 2.0 * Ip, np.zeros((nP, p)))
    if nL:
        Il = np.tile(np.eye(l).reshape(-1), (nL, 1))
        o.set_edge_data(kl, Il, None, 2.0 * Il, np.zeros((nL, l)))
    return o


@gpu
@pytest.mark.parametrize("huber", [0.0, 0.8])
@pytest.mark.parametrize("with_landmark", [False, True])
def test_ternary_edges_as_three_pair_sets_equal_the_oracle_and_the_dense_quadratic_form(huber, with_landmark):
    T = ternary_problem(with_landmark)
    p, l, d, nP, nL, n, v, dims, offs, J, omega, err, rng = (T[k] for k in ("p", "l", "d", "nP", "nL", "n", "v", "dims", "offs", "J", "omega", "err", "rng"))
    # every free vertex gets a unary prior so that the system is positive definite whatever the random topology
    s = capi.HipBlockSolver(p, l, 0)
    # vertex 0 and 1: their own terms (diagonal block, right-hand side) and chi2 from the pair (0, 1); vertex 2: from the pair (1, 2)
    pairs = [(0, 1, 0), (0, 2, capi.PART_NO_VERTEX0 | capi.PART_NO_VERTEX1 | capi.PART_NO_CHI2), (1, 2, capi.PART_NO_VERTEX0 | capi.PART_NO_CHI2)]
    ids = []
    for i, j, parts in pairs:
        k = s.addEdgeSet(d, v[:, i], v[:, j])
        s.setEdgeSetParts(k, parts)
        ids.append(k)
    # priors: one unary set per vertex class
    prior_p = s.addEdgeSet(p, np.arange(nP))
    prior_l = s.addEdgeSet(l, np.arange(nL) + nP) if nL else None
    s.buildStructure(nP, nL, nL > 0)
    for (i, j, parts), k in zip(pairs, ids):
        s.setEdgeData(k, col(J[i]), col(J[j]), col(omega), err)
        if huber > 0:
            s.setRobustKernel(k, capi.KERNEL_HUBER, huber)
    Ip = np.tile(np.eye(p).reshape(-1), (nP, 1))
    s.setEdgeData(prior_p, Ip, None, 2.0 * Ip, np.zeros((nP, p)))
    if nL:
        Il = np.tile(np.eye(l).reshape(-1), (nL, 1))
        s.setEdgeData(prior_l, Il, None, 2.0 * Il, np.zeros((nL, l)))
    s.buildSystem()
    H, b, chi = _dense(int(dims.sum()), dims, offs, v, J, omega, err, huber)
    H += 2.0 * np.eye(len(b))                                   # the priors (zero error: nothing in b or chi2)
    assert abs(s.chi2() - chi) <= 1e-12 * chi
    assert np.abs(s.b() - b).max() <= 1e-12 * np.abs(b).max()
    # H through its action on random vectors (dest += H src, full system)
    for _ in range(3):
        x = rng.normal(size=len(b))
        y = s.multiplyHessian(x)
        assert np.abs(y - H @ x).max() <= 1e-11 * np.abs(H @ x).max()
    # the oracle's BaseMultiEdge restatement: block by block
    o = oracle_ternary(T, huber)
    o.build_system()
    assert abs(s.chi2() - o.chi2()) <= 1e-12 * o.chi2()
    assert np.abs(s.b() - o.b()).max() <= 1e-12 * np.abs(o.b()).max()
    for which, name in ((capi.HPP, "Hpp"),) + (((capi.HPL, "Hpl"), (capi.HLL, "Hll")) if nL else ()):
        ref = o.values(name)
        assert np.abs(s.values(which) - ref).max() <= 1e-12 * np.abs(ref).max(), name
    assert s.solve() and o.solve()
    xs = np.linalg.solve(H, b)
    assert np.abs(s.x() - xs).max() <= 1e-9 * np.abs(xs).max()
    assert np.abs(s.x() - o.x()).max() <= 1e-9 * np.abs(o.x()).max()


@gpu
def test_parts_are_validated():
    s = capi.HipBlockSolver(3, 2, 0)
    k = s.addEdgeSet(2, np.array([0, 1]), np.array([1, 2]))
    with pytest.raises(Exception):
        s.setEdgeSetParts(k, 8)
    with pytest.raises(Exception):
        s.setEdgeSetParts(k + 1, 1)
    u = s.addEdgeSet(3, np.array([0, 1, 2]))
    with pytest.raises(Exception):
        s.setEdgeSetParts(u, capi.PART_NO_VERTEX1)
