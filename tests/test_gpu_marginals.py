"""computeMarginals / solvePattern: blocks of the inverse of the pose system from the device
factorisation against the CPU oracle (unit right-hand sides through the restated CSparse path) and, for BA with the
Schur complement, against the pose block of the dense inverse of the full Hessian."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import ba_case, hip_ba, manhattan_golden, oracle_ba, relerr

pytestmark = pytest.mark.gpu


def test_pose_graph_marginals_match_oracle_solves():
    from openslam_g2o_amd import capi
    g = manhattan_golden()
    J0, J1, err = O.se2_edges(g["estimates"], g["vi"], g["vj"], g["meas"])
    s = capi.HipBlockSolver(3, 2, 0)
    k = s.addEdgeSet(3, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    s.buildStructure(g["nP"], 0, False)
    s.setEdgeData(k, J0, J1, g["omega"], err)
    s.buildSystem()
    rows = np.array([0, 5, 5, 3498, 1200, 7], np.int32)
    cols = np.array([0, 5, 7, 3498, 3000, 5], np.int32)
    M = s.computeMarginals(rows, cols)
    assert M is not None and M.shape == (6, 3, 3)
    o = O.OracleSolver(3, 2, g["nP"], 0, schur=False)
    ko = o.add_edge_set(3, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    o.set_dims(ko, 3, 3)
    o.build_structure()
    o.set_edge_data(ko, J0, J1, g["omega"], err)
    o.build_system()
    cp, row = o.pattern("pp")
    val = o.values("Hpp")
    n = 3 * g["nP"]
    for c in np.unique(cols):
        for kk in range(3):
            e = np.zeros(n)
            e[3 * c + kk] = 1.0
            ok, x, _ = O.linear_solve_blocks(g["nP"], 3, cp, row, val, e)
            assert ok
            for i in np.flatnonzero(cols == c):
                r = rows[i]
                assert np.abs(M[i][:, kk] - x[3 * r:3 * r + 3]).max() <= 1e-9 * np.abs(x).max()
    assert np.abs(M[2] - M[5].T).max() <= 1e-9 * np.abs(M[2]).max()      # (5,7) and (7,5): the inverse is symmetric
    assert np.all(np.linalg.eigvalsh(0.5 * (M[0] + M[0].T)) > 0)         # a covariance block


def test_ba_marginals_default_is_the_inverse_of_hpp_like_the_reference():
    """BlockSolver::computeMarginals hands *_Hpp to solvePattern also under Schur (block_solver.hpp:489-493): the
    default result is the inverse of Hpp alone (dense numpy inverse of the oracle's Hpp)."""
    pr = ba_case(10, 40)
    s = hip_ba(pr)
    s.buildSystem()
    o = oracle_ba(pr)
    o.build_system()
    nP = pr["nP"]
    Hpp = o.dense_full()[:6 * nP, :6 * nP]
    Hinv = np.linalg.inv(Hpp)
    rows = np.array([0, 2, 7, 3], np.int32)
    cols = np.array([0, 5, 7, 1], np.int32)
    M = s.computeMarginals(rows, cols)
    for i in range(len(rows)):
        ref = Hinv[6 * rows[i]:6 * rows[i] + 6, 6 * cols[i]:6 * cols[i] + 6]
        assert np.abs(M[i] - ref).max() <= 1e-9 * np.abs(Hinv).max()
    s.setLambda(1.0, True)      # the reduced system is formed again by the next solve
    assert s.solve()
    xo = None
    o.set_lambda(1.0, True)
    assert o.solve()
    assert relerr(s.x(), o.x()) < 1e-7


def test_ba_pose_marginals_are_the_pose_block_of_the_full_inverse():
    pr = ba_case(10, 40)
    s = hip_ba(pr)
    s.setOption("marginals_reduced", 1)
    s.buildSystem()
    o = oracle_ba(pr)
    o.build_system()
    H = o.dense_full()
    nP = pr["nP"]
    Hinv = np.linalg.inv(H)
    rows = np.array([0, 2, 7, 3], np.int32)
    cols = np.array([0, 5, 7, 1], np.int32)
    M = s.computeMarginals(rows, cols)
    for i in range(len(rows)):
        ref = Hinv[6 * rows[i]:6 * rows[i] + 6, 6 * cols[i]:6 * cols[i] + 6]
        assert np.abs(M[i] - ref).max() <= 1e-7 * np.abs(Hinv[:6 * nP, :6 * nP]).max()
    # the solver still solves afterwards
    s.setLambda(1.0, True)
    assert s.solve()


def test_sparse_inverse_gives_every_block_of_the_pattern():
    """computeMarginals by the sparse-inverse recursion over the frontal matrices (one top-down pass for ALL requested
    blocks): every block of the reduced pattern of a BA system -- diagonal and off-diagonal, both orientations --
    against the dense inverse of Hpp, and against the column-by-column path (option marginals_recursion = 0)."""
    from openslam_g2o_amd import capi
    pr = ba_case(90, 400)
    s = hip_ba(pr)
    s.buildSystem()
    o = oracle_ba(pr)
    o.build_system()
    nP = pr["nP"]
    Hinv = np.linalg.inv(o.dense_full()[:6 * nP, :6 * nP])
    cp, ri = s.pattern(capi.HSCHUR)
    rows, cols = [], []
    for c in range(nP):
        for q in range(cp[c], cp[c + 1]):
            rows += [ri[q], c]
            cols += [c, ri[q]]
    rows, cols = np.array(rows, np.int32), np.array(cols, np.int32)
    M = s.computeMarginals(rows, cols)
    assert M is not None and M.shape == (len(rows), 6, 6)
    scale = np.abs(Hinv).max()
    for i in range(len(rows)):
        ref = Hinv[6 * rows[i]:6 * rows[i] + 6, 6 * cols[i]:6 * cols[i] + 6]
        assert np.abs(M[i] - ref).max() <= 1e-9 * scale, (rows[i], cols[i])
    s.setOption("marginals_recursion", 0)
    sel = np.arange(0, len(rows), 37)
    M0 = s.computeMarginals(rows[sel], cols[sel])
    assert np.abs(M0 - M[sel]).max() <= 1e-9 * scale
    s.setLambda(1.0, True)      # the solver still solves afterwards
    assert s.solve()


def test_sparse_inverse_with_large_fronts_matches_column_solves():
    """Pose graph with fronts of several hundred rows (scratch-slab fronts: their panels come from the whole-GPU passes):
    diagonal blocks spread over the tree and a few coupled pairs, recursion against unit right-hand sides; a pair outside
    the pattern of the factor falls back to the column path inside the same call."""
    from openslam_g2o_amd import capi
    from tests.helpers import sphere_golden
    g = sphere_golden()
    J0, J1, err = O.se3_edges(g["poses"], g["vi"], g["vj"], g["Z"])
    s = capi.HipBlockSolver(6, 3, 0)
    k = s.addEdgeSet(6, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    s.buildStructure(g["nP"], 0, False)
    s.setEdgeData(k, J0, J1, g["omega"], err)
    s.buildSystem()
    s.setLambda(1e-3 * s.maxDiagonal(), True)
    nP = g["nP"]
    diag = np.arange(0, nP, 97, dtype=np.int32)
    hi, hj = g["hidx"][g["vi"]], g["hidx"][g["vj"]]
    keep = (hi >= 0) & (hj >= 0)
    pi, pj = hi[keep][::701].astype(np.int32), hj[keep][::701].astype(np.int32)   # measured pairs: blocks of the pattern
    rows = np.concatenate([diag, pi, [0]]).astype(np.int32)
    cols = np.concatenate([diag, pj, [nP - 1]]).astype(np.int32)
    M = s.computeMarginals(rows, cols)
    s.setOption("marginals_recursion", 0)
    M0 = s.computeMarginals(rows, cols)
    assert M is not None and M0 is not None
    assert np.abs(M - M0).max() <= 1e-9 * np.abs(M0).max()
