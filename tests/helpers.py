"""Shared builders for the parity tests: the same seeded inputs are fed to the HIP path
(through the C ABI) and to the CPU oracle."""
import os

import numpy as np

from openslam_g2o_amd import g2o_io, synthetic as S
from oracle import oracle as O

TOL_DX = 1e-8          # stated fp64 tolerance on dx where the system is well conditioned

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ba_case(P, L, seed=42, outlier_frac=0.0, obs_per_landmark=5):
    pr = S.make_ba_problem(P, L, seed=seed, outlier_frac=outlier_frac, obs_per_landmark=obs_per_landmark)
    Jp, Jc, err = S.ba_linearize(pr)
    pr.update(Jp=Jp, Jc=Jc, err=err, omega=S.ba_omega(pr))
    return pr


def oracle_ba(pr, huber=0.0, schur=True):
    o = O.OracleSolver(6, 3, pr["nP"], pr["nL"], schur)
    k = o.add_edge_set(2, pr["v0"], pr["v1"])
    o.set_dims(k, 3, 6)
    o.build_structure()
    o.set_edge_data(k, pr["Jp"], pr["Jc"], pr["omega"], pr["err"], huber)
    return o


def hip_ba(pr, huber=0.0, schur=True, device=0, options=None):
    from openslam_g2o_amd import capi
    s = capi.HipBlockSolver(6, 3, device)
    for name, value in (options or {}).items():      # analysis-time knobs go in before buildStructure
        s.setOption(name, value)
    k = s.addEdgeSet(2, pr["v0"], pr["v1"])
    s.buildStructure(pr["nP"], pr["nL"], schur)
    s.setEdgeData(k, pr["Jp"], pr["Jc"], pr["omega"], pr["err"])
    if huber > 0:
        s.setRobustKernel(k, capi.KERNEL_HUBER, huber)
    return s


def manhattan_golden():
    g = dict(np.load(os.path.join(GOLD, "manhattan3500.npz")))
    nv = len(g["estimates"])
    h, nP = g2o_io.hessian_index(nv, [0])
    info = np.zeros((len(g["vi"]), 3, 3))
    k = 0
    for i in range(3):
        for j in range(i, 3):
            info[:, i, j] = info[:, j, i] = g["info_upper"][:, k]
            k += 1
    g["omega"] = info.transpose(0, 2, 1).reshape(-1, 9).copy()
    g["hidx"] = h
    g["nP"] = nP
    return g


def relerr(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def sphere_golden():
    """Config 2 fixture (sphere, VertexSE3/EdgeSE3): parse results + reference CSparse answers."""
    g = dict(np.load(os.path.join(GOLD, "sphere2200.npz")))
    nv = len(g["estimates"])
    h, nP = g2o_io.hessian_index(nv, [0])
    ne = len(g["vi"])
    info = np.zeros((ne, 6, 6))
    k = 0
    for i in range(6):
        for j in range(i, 6):
            info[:, i, j] = info[:, j, i] = g["info_upper"][:, k]
            k += 1
    g["omega"] = info.transpose(0, 2, 1).reshape(-1, 36).copy()
    g["hidx"], g["nP"] = h, nP
    g["poses"] = O.se3_from_qt(g["estimates"], normalize=False)
    g["Z"] = O.se3_from_qt(g["meas"], normalize=True)
    return g


def dx_tolerance(o):
    """max(TOL_DX, 4 cond eps) with cond of the oracle's (damped) reduced pose system, dense."""
    cp, ri = o.pattern("hs")
    p = o.p
    nb = len(cp) - 1
    H = np.zeros((nb * p, nb * p))
    V = o.values("Hschur").reshape(-1, p, p)
    for c in range(nb):
        for q in range(cp[c], cp[c + 1]):
            r = ri[q]
            blk = V[q].T                                  # column-major block
            H[r * p:(r + 1) * p, c * p:(c + 1) * p] = blk
            H[c * p:(c + 1) * p, r * p:(r + 1) * p] = blk.T
    ev = np.linalg.eigvalsh(H)
    cond = ev[-1] / ev[0]
    return max(TOL_DX, 4.0 * cond * np.finfo(float).eps), cond
