"""CPU-only: pins the oracle (the CPU restatement of the reference path) against
 (a) the committed golden vectors produced by the REFERENCE's compiled CSparse path,
 (b) that reference path itself when oracle/_ref is present (bit-exact under equal ordering),
 (c) dense numpy solves and the Schur == full-system identity."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import ba_case, manhattan_golden, oracle_ba, relerr, GOLD
import os


def _manhattan_system(g):
    J0, J1, err = O.se2_edges(g["estimates"], g["vi"], g["vj"], g["meas"])
    s = O.OracleSolver(3, 2, g["nP"], 0, schur=False)
    k = s.add_edge_set(3, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    s.set_dims(k, 3, 3)
    s.build_structure()
    s.set_edge_data(k, J0, J1, g["omega"], err)
    s.build_system()
    return s


def test_manhattan_golden_reference_csparse():
    g = manhattan_golden()
    s = _manhattan_system(g)
    assert s.pattern("pp")[1].size == int(g["nnzb"]) == 8949
    assert abs(s.chi2() - g["chi2_gn"][0]) <= 1e-12 * g["chi2_gn"][0]
    np.testing.assert_allclose(s.b(), g["b0"], rtol=0, atol=1e-9 * np.abs(g["b0"]).max())
    # own ordering: same solution as the reference's to roundoff
    assert s.solve()
    assert relerr(s.x(), g["x_gn0"]) < 1e-9
    # the reference's block AMD ordering: bit-exact (same loop structure as cs_chol_workspace)
    s.set_ordering(2, g["block_perm"])
    assert s.solve()
    assert np.array_equal(s.x(), g["x_gn0"])
    assert s.lnz() == float(g["lnz_block_amd"])
    # LM-damped system
    s.set_lambda(float(g["lambda0"]), True)
    assert s.solve()
    assert np.array_equal(s.x(), g["x_lm0"])
    s.restore_diagonal()
    assert s.solve()
    assert np.array_equal(s.x(), g["x_gn0"])                             # restoreDiagonal is exact


def test_manhattan_gn_trajectory():
    g = manhattan_golden()
    est = g["estimates"].copy()
    for it in range(5):
        J0, J1, err = O.se2_edges(est, g["vi"], g["vj"], g["meas"])
        s = O.OracleSolver(3, 2, g["nP"], 0, schur=False)
        k = s.add_edge_set(3, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
        s.set_dims(k, 3, 3)
        s.build_structure()
        s.set_edge_data(k, J0, J1, g["omega"], err)
        s.build_system()
        assert abs(s.chi2() - g["chi2_gn"][it]) <= 1e-7 * g["chi2_gn"][it]
        assert s.solve()
        est = O.se2_oplus(est, g["hidx"], s.x())
    assert abs(g["chi2_gn"][-1] - 146.0766) < 1e-3      # the well-known manhattan3500 optimum


def test_bitwise_against_reference_build():
    if O.ref() is None:      # (looked up here, not at import: a -m gpu collection must not load oracle/_ref)
        pytest.skip("oracle/_ref (reference CSparse build) not present")
    g = manhattan_golden()
    s = _manhattan_system(g)
    cp, row = s.pattern("pp")
    ok, x, lnz, P = O.ref_solve_blocks(g["nP"], 3, cp, row, s.values("Hpp"), s.b())
    assert ok and np.array_equal(P, g["block_perm"]) and np.array_equal(x, g["x_gn0"])
    s.set_ordering(2, P)
    assert s.solve() and np.array_equal(s.x(), x) and s.lnz() == lnz


def test_ba_small_golden_and_schur_identity():
    gold = dict(np.load(os.path.join(GOLD, "ba_small.npz")))
    pr = ba_case(20, 200)
    assert abs(float(np.sum(pr["meas"])) - gold["meas_checksum"]) < 1e-6      # generator determinism
    o = oracle_ba(pr)
    o.build_system()
    assert abs(o.chi2() - gold["chi2"]) <= 1e-12 * gold["chi2"]
    np.testing.assert_allclose(o.b(), gold["b"], rtol=0, atol=1e-12 * np.abs(gold["b"]).max())
    o.set_lambda(float(gold["lam"]), True)
    assert o.solve()
    x = o.x()
    assert relerr(x, gold["x_dense"]) < 1e-9                                   # Schur path == dense full solve
    assert relerr(x[:6 * pr["nP"]], gold["xp_ref_csparse"]) < 1e-9            # == reference CSparse on Hschur
    assert relerr(o.values("Hschur"), gold["Hschur"]) < 1e-13
    # Schur path == full-system (no Schur) sparse solve
    o2 = oracle_ba(pr, schur=False)
    # without Schur the landmark blocks cannot be expressed in BlockSolver<6,3>'s pose matrix;
    # use the dense full system instead
    H = o.dense_full()
    assert relerr(np.linalg.solve(H, o.b()), x) < 1e-9
    # residual through multiply_full
    r = o.multiply_full(x) - o.b()
    assert np.abs(r).max() <= 1e-10 * np.abs(o.b()).max()
    del o2


def test_huber_weights_and_not_pd():
    pr = ba_case(12, 80, outlier_frac=0.2)
    o = oracle_ba(pr, huber=1.0)
    o.build_system()
    # robust chi2 <= plain chi2, and equals the closed form
    e2 = np.sum(pr["err"] ** 2, axis=1)
    rho = np.where(e2 <= 1.0, e2, 2 * np.sqrt(e2) - 1.0)
    assert abs(o.chi2() - rho.sum()) <= 1e-12 * rho.sum()
    o.set_lambda(1.0, True)
    assert o.solve()
    o.restore_diagonal()
    # strongly negative damping makes the system indefinite -> solve() == false (csparse_helper.cpp:136)
    o.set_lambda(-10.0 * o.max_diagonal(), True)
    assert not o.solve()


def test_sphere_golden_reference_csparse():
    """Config 2 (sphere, BlockSolver_6_3 without Schur): oracle == reference CSparse golden vector."""
    from tests.helpers import sphere_golden
    g = sphere_golden()
    J0, J1, err = O.se3_edges(g["poses"], g["vi"], g["vj"], g["Z"])
    s = O.OracleSolver(6, 3, g["nP"], 0, schur=False)
    k = s.add_edge_set(6, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    s.set_dims(k, 6, 6)
    s.build_structure()
    s.set_edge_data(k, J0, J1, g["omega"], err)
    s.build_system()
    assert s.pattern("pp")[1].size == int(g["nnzb"]) == 10843
    assert abs(s.chi2() - g["chi2_lm"][0]) <= 1e-12 * g["chi2_lm"][0]
    s.set_ordering(2, g["block_perm"])
    assert s.solve()
    assert np.array_equal(s.x(), g["x_gn0"]) and s.lnz() == float(g["lnz_block_amd"])
    s.set_lambda(float(g["lambda0"]), True)
    assert s.solve()
    assert np.array_equal(s.x(), g["x_lm0"])


def test_se3_jacobian_against_central_differences():
    """The reference's own check (g2o/types/slam3d/test_slam3d_jacobian.cpp:116-148): analytic
    EdgeSE3 Jacobian vs central differences, tolerance 1e-6."""
    rng = np.random.default_rng(0)
    n = 100

    def rand_iso(n):
        q = rng.normal(size=(n, 4))
        q /= np.linalg.norm(q, axis=1)[:, None]
        return O.se3_from_qt(np.hstack([rng.normal(size=(n, 3)), q]))
    poses = np.vstack([rand_iso(n), rand_iso(n)])
    Z = rand_iso(n)
    vi = np.arange(n, dtype=np.int32)
    vj = vi + n
    J0, J1, _ = O.se3_edges(poses, vi, vj, Z)
    hid = np.arange(2 * n, dtype=np.int32)
    h = 1e-6
    for side, J in ((0, J0), (1, J1)):
        for c in range(6):
            x = np.zeros((2 * n, 6))
            x[(vi if side == 0 else vj), c] = h
            ep = O.se3_edges(O.se3_oplus(poses, hid, x.reshape(-1)), vi, vj, Z, jac=False)
            em = O.se3_edges(O.se3_oplus(poses, hid, -x.reshape(-1)), vi, vj, Z, jac=False)
            assert np.abs((ep - em) / (2 * h) - J.reshape(n, 6, 6)[:, c, :]).max() < 1e-6


def test_robust_kernels_known_values_and_consistency():
    """The five robust kernels of the reference (robust_kernel_impl.cpp:65-126) in the oracle: closed-form values at
    hand-computed points, rho' = d rho / de and rho'' = d rho' / de by central differences, no-op below the threshold."""
    import math
    d = 1.5
    # Huber: inlier identity, outlier 2 d sqrt(e) - d^2
    assert np.allclose(O.robustify(1, d, 1.0), [1.0, 1.0, 0.0])
    assert np.allclose(O.robustify(1, d, 9.0), [2 * 3 * d - d * d, d / 3.0, -0.5 * (d / 3.0) / 9.0])
    # PseudoHuber / Cauchy at e = 3 d^2: aux = 4
    e = 3 * d * d
    assert np.allclose(O.robustify(2, d, e), [2 * d * d * (2 - 1), 0.5, -0.5 / (d * d) * 0.5 / 4])
    assert np.allclose(O.robustify(3, d, e), [d * d * math.log(4.0), 0.25, -(1 / (d * d)) * 0.0625])
    # Saturated: clamps at d^2
    assert np.allclose(O.robustify(4, d, 1.0), [1.0, 1.0, 0.0]) and np.allclose(O.robustify(4, d, 10.0), [d * d, 0.0, 0.0])
    # DCS (delta = phi): scale = 2 phi / (phi + e), capped at 1
    assert np.allclose(O.robustify(5, d, 1.0), [1.0, 1.0, 0.0])
    sc = 2 * d / (d + 6.0)
    assert np.allclose(O.robustify(5, d, 6.0), [sc * sc * 6.0, sc * sc, 0.0])
    for kind in (1, 2, 3):
        for e in (0.3, 4.0, 50.0):
            h = 1e-6 * max(1.0, e)
            r, rp, rm = O.robustify(kind, d, e), O.robustify(kind, d, e + h), O.robustify(kind, d, e - h)
            assert abs((rp[0] - rm[0]) / (2 * h) - r[1]) < 1e-6
            assert abs((rp[1] - rm[1]) / (2 * h) - r[2]) < 1e-6


def test_ba_projection_jacobian_against_central_differences():
    """EdgeProjectXYZ2UV (types_six_dof_expmap.cpp:288-326) in the oracle, pinned independently of both builder-written
    producers: the analytic Jacobians against central differences of a projection written here, with the pose update
    taken as the matrix exponential of the twist (scipy.linalg.expm) -- VertexSE3Expmap::oplusImpl is
    exp(update) * estimate with update = (omega, upsilon) (types_six_dof_expmap.h:101-104, se3quat.h:223-257) -- and the
    point update as plain addition (VertexSBAPointXYZ).  Also pins the oracle's own oplus against the same exponential.
    Tolerance 1e-6 like the reference's Jacobian test (types/slam3d/test_slam3d_jacobian.cpp)."""
    from scipy.linalg import expm
    rng = np.random.default_rng(3)
    n = 60
    f, cx, cy = 1000.0, 320.0, 240.0
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    iso = O.se3_from_qt(np.hstack([0.3 * rng.normal(size=(n, 3)), q]))           # [n][12]: R column-major | t
    pts = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(3, 5, n)], axis=1)
    # points in front of every camera: express them in the camera frame, map back to the world
    R = iso[:, :9].reshape(n, 3, 3).transpose(0, 2, 1)
    t = iso[:, 9:]
    world = np.einsum("nji,nj->ni", R, pts - t)                                   # X = R' (Xc - t)
    meas = rng.uniform(0, 640, size=(n, 2))
    idx = np.arange(n, dtype=np.int32)

    def project(Rm, tv, X):
        Xc = np.einsum("nij,nj->ni", Rm, X) + tv
        return np.stack([meas[:, 0] - (Xc[:, 0] / Xc[:, 2] * f + cx), meas[:, 1] - (Xc[:, 1] / Xc[:, 2] * f + cy)], axis=1)

    def hat(d):                      # twist (omega, upsilon) -> 4 x 4
        w, u = d[:3], d[3:]
        return np.array([[0, -w[2], w[1], u[0]], [w[2], 0, -w[0], u[1]], [-w[1], w[0], 0, u[2]], [0, 0, 0, 0.0]])

    def moved(d):
        E = expm(hat(d))
        return np.einsum("ij,njk->nik", E[:3, :3], R), (E[:3, :3] @ t.T).T + E[:3, 3]

    Jp, Jc, err = O.ba_edges(iso, world, idx, idx, meas, f, cx, cy)
    assert np.abs(err - project(R, t, world)).max() < 1e-9
    h = 1e-6
    for c in range(6):
        d = np.zeros(6)
        d[c] = h
        Rp, tp = moved(d)
        Rm, tm = moved(-d)
        num = (project(Rp, tp, world) - project(Rm, tm, world)) / (2 * h)
        assert np.abs(num - Jc.reshape(n, 6, 2)[:, c, :]).max() < 1e-6 * max(1.0, np.abs(Jc).max())
    for c in range(3):
        d = np.zeros(3)
        d[c] = h
        num = (project(R, t, world + d) - project(R, t, world - d)) / (2 * h)
        assert np.abs(num - Jp.reshape(n, 3, 2)[:, c, :]).max() < 1e-6 * max(1.0, np.abs(Jp).max())
    # the oracle's pose update is the same exponential
    x = 0.05 * rng.normal(size=(n, 6))
    cams2, _ = O.ba_oplus(iso, world, idx, idx, np.concatenate([x.reshape(-1), np.zeros(3 * n)]), 6 * n)
    for k in range(0, n, 7):
        E = expm(hat(x[k]))
        Rk = E[:3, :3] @ R[k]
        tk = E[:3, :3] @ t[k] + E[:3, 3]
        assert np.abs(cams2[k, :9].reshape(3, 3).T - Rk).max() < 1e-12 and np.abs(cams2[k, 9:] - tk).max() < 1e-12


@pytest.mark.parametrize("huber", [0.0, 0.8])
@pytest.mark.parametrize("with_landmark", [False, True])
def test_n_ary_edges_of_the_oracle_equal_the_dense_quadratic_form(huber, with_landmark):
    """BaseMultiEdge::constructQuadraticForm as the oracle restates it (orc_add_multi_edge_set: the computeUpperTriangleIndex
    block table and the transposed helper blocks of base_multi_edge.hpp:128-222 over block_solver.hpp:208-251) against a dense
    NumPy assembly of the same quadratic form: H through every stored block, b, chi2 and the solution."""
    from tests.test_gpu_multi_edge import _dense, oracle_ternary, ternary_problem
    T = ternary_problem(with_landmark)
    o = oracle_ternary(T, huber)
    o.build_system()
    n_tot = int(T["dims"].sum())
    H, b, chi = _dense(n_tot, T["dims"], T["offs"], T["v"], T["J"], T["omega"], T["err"], huber)
    H += 2.0 * np.eye(n_tot)
    assert abs(o.chi2() - chi) <= 1e-12 * chi
    assert np.abs(o.b() - b).max() <= 1e-12 * np.abs(b).max()
    p, l, nP, nL = T["p"], T["l"], T["nP"], T["nL"]
    Ho = np.zeros_like(H)
    cp, ri = o.pattern("pp")
    V = o.values("Hpp").reshape(-1, p, p)
    for c in range(nP):
        for q in range(cp[c], cp[c + 1]):
            r = ri[q]
            Ho[r * p:(r + 1) * p, c * p:(c + 1) * p] = V[q].T            # (column-major blocks)
            Ho[c * p:(c + 1) * p, r * p:(r + 1) * p] = V[q]
    if nL:
        cp, ri = o.pattern("pl")
        V = o.values("Hpl").reshape(-1, l, p)
        D = o.values("Hll").reshape(-1, l, l)
        for c in range(nL):
            a = nP * p + c * l
            Ho[a:a + l, a:a + l] = D[c].T
            for q in range(cp[c], cp[c + 1]):
                r = ri[q]
                Ho[r * p:(r + 1) * p, a:a + l] = V[q].T
                Ho[a:a + l, r * p:(r + 1) * p] = V[q]
    assert np.abs(Ho - H).max() <= 1e-12 * np.abs(H).max()
    assert o.solve()
    xs = np.linalg.solve(H, b)
    assert np.abs(o.x() - xs).max() <= 1e-9 * np.abs(xs).max()
