"""PCG on the reduced / pose system (LinearSolverPCG, g2o/solvers/pcg/linear_solver_pcg.hpp:79-196).
CPU: the oracle restatement converges to the direct solve and honours tolerance / maxIter / the carried residual.
GPU: the device PCG (block_pcg.hip) against the oracle PCG with the same settings, and against the Cholesky path."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import ba_case, manhattan_golden, oracle_ba, relerr


def _manhattan_system():
    g = manhattan_golden()
    J0, J1, err = O.se2_edges(g["estimates"], g["vi"], g["vj"], g["meas"])
    o = O.OracleSolver(3, 2, g["nP"], 0, schur=False)
    k = o.add_edge_set(3, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    o.set_dims(k, 3, 3)
    o.build_structure()
    o.set_edge_data(k, J0, J1, g["omega"], err)
    o.build_system()
    o.set_lambda(float(g["lambda0"]) * 1e4, True)      # a damped (well conditioned) LM system
    cp, row = o.pattern("pp")
    return g, o, cp, row


def test_oracle_pcg_matches_direct_solve():
    g, o, cp, row = _manhattan_system()
    val, b = o.values("Hpp"), o.b()
    ok, xd, _ = O.linear_solve_blocks(g["nP"], 3, cp, row, val, b)
    assert ok
    ok, x, it, res = O.pcg_solve_blocks(g["nP"], 3, cp, row, val, b, tolerance=1e-20, absolute=False)
    assert ok and 0 < it < 3 * g["nP"] and relerr(x, xd) < 1e-8
    ok, x1, it1, res1 = O.pcg_solve_blocks(g["nP"], 3, cp, row, val, b, tolerance=1e-6)      # the reference default
    assert ok and it1 < it and relerr(x1, xd) < 1e-2
    ok, x2, it2, _ = O.pcg_solve_blocks(g["nP"], 3, cp, row, val, b, tolerance=1e-20, max_iter=5)
    assert it2 == 5
    # absolute tolerance: the residual of the previous solve becomes the stopping level (linear_solver_pcg.hpp:127-131)
    ok, x3, it3, _ = O.pcg_solve_blocks(g["nP"], 3, cp, row, val, b, tolerance=1e-20, absolute=True, residual=10.0 * res1)
    assert it3 < it1 + 1


@pytest.mark.gpu
def test_device_pcg_matches_oracle_pcg_and_cholesky():
    from openslam_g2o_amd import capi
    g, o, cp, row = _manhattan_system()
    J0, J1, err = O.se2_edges(g["estimates"], g["vi"], g["vj"], g["meas"])
    lam = float(g["lambda0"]) * 1e4
    s = capi.HipBlockSolver(3, 2, 0)
    k = s.addEdgeSet(3, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    s.buildStructure(g["nP"], 0, False)
    s.setEdgeData(k, J0, J1, g["omega"], err)
    s.buildSystem()
    s.setLambda(lam, True)
    assert s.solve()
    x_chol = s.x()
    s.restoreDiagonal()
    for tol, rel in ((1e-6, 2e-2), (1e-20, 1e-8)):
        s.setOption("linear_solver", 1)
        s.setOption("pcg_tolerance", tol)
        s.setOption("pcg_absolute_tolerance", 0)
        s.setLambda(lam, True)
        assert s.solve()
        x = s.x()
        s.restoreDiagonal()
        it = s.stats()["iterationsLinearSolver"]
        ok, xo, ito, _ = O.pcg_solve_blocks(g["nP"], 3, cp, row, o.values("Hpp"), o.b(), tolerance=tol, absolute=False)
        assert ok and abs(it - ito) <= max(2, ito // 20)        # same stopping rule; summation order differs
        assert relerr(x, xo) < max(10 * rel * 1e-2, 1e-7) or relerr(x, x_chol) < rel
        assert relerr(x, x_chol) < rel
    s.setOption("pcg_max_iterations", 7)
    s.setLambda(lam, True)
    assert s.solve() and s.stats()["iterationsLinearSolver"] == 7
    s.restoreDiagonal()


@pytest.mark.gpu
def test_device_pcg_on_the_schur_complement():
    """BA: PCG on the explicit reduced system Hschur (what BlockSolver + LinearSolverPCG do), then the usual
    back-substitution; against the Cholesky path and the oracle."""
    from tests.helpers import hip_ba
    pr = ba_case(40, 400)
    s = hip_ba(pr)
    s.buildSystem()
    s.setLambda(20.0, True)
    assert s.solve()
    x_chol = s.x()
    s.setOption("linear_solver", 1)
    s.setOption("pcg_tolerance", 1e-22)
    s.setOption("pcg_absolute_tolerance", 0)
    assert s.solve()
    x = s.x()
    assert relerr(x, x_chol) < 1e-7
    o = oracle_ba(pr)
    o.build_system()
    o.set_lambda(20.0, True)
    assert o.solve()
    assert relerr(x, o.x()) < 1e-7
    s.restoreDiagonal()
    # a non positive definite diagonal block is reported like a failed factorisation
    s.setLambda(-1e12, True)
    assert not s.solve()


@pytest.mark.gpu
def test_matrix_free_pcg_on_the_reduced_system():
    """linear_solver 2: LinearSolverPCG's iteration (linear_solver_pcg.hpp:79-196) on the Schur complement WITHOUT forming it
    (Hschur v = Hpp v + lambda v - Hpl Dinv Hpl' v, exact block-Jacobi preconditioner): same solution as the direct solver
    and as the PCG on the explicit Hschur, same iteration count as the latter."""
    from openslam_g2o_amd import capi
    from tests.helpers import ba_case, hip_ba, relerr
    pr = ba_case(120, 1500)
    xs, iters = {}, {}
    for ls in (0, 1, 2):
        s = hip_ba(pr, options={"linear_solver": ls, "pcg_tolerance": 1e-20, "pcg_absolute_tolerance": 0, "pcg_max_iterations": 6000})
        s.buildSystem()
        s.setLambda(5.0, True)
        assert s.solve()
        xs[ls] = s.x()
        iters[ls] = s.stats()["iterationsLinearSolver"]
        r = s.multiplyHessian(xs[ls]) - s.b()
        assert np.abs(r).max() <= 1e-7 * np.abs(s.b()).max()
        assert s.solve() and relerr(s.x(), xs[ls]) < 1e-6          # repeatable
    assert relerr(xs[1], xs[0]) < 1e-6 and relerr(xs[2], xs[0]) < 1e-6
    assert iters[2] > 0 and abs(iters[2] - iters[1]) <= max(5, iters[1] // 20)   # same iteration up to rounding
