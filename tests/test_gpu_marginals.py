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
