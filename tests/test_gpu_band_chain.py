"""The sliding-window kernel for leaf chains of a band (openslam_g2o_amd/csrc/band_chain.inc) against the oracle and
against the general register-tile kernel it replaces there (option band_kernel = 0): same L panels, same update
matrices, hence the same solution up to the order of the floating-point sums.  Both sources of the reduced system are
covered (the Schur reduction folded into the front assembly, and a materialised Hschur), chains that end in the middle of a 16-column tile, Huber weights, and a graph that is NOT a band (nothing qualifies)."""
import numpy as np
import pytest

from tests.helpers import ba_case, dx_tolerance, hip_ba, oracle_ba, relerr

pytestmark = pytest.mark.gpu


def _solve(pr, lam, options, huber=0.0):
    s = hip_ba(pr, huber=huber, options=options)
    s.buildSystem()
    s.setLambda(lam, True)
    ok = s.solve()
    s.restoreDiagonal()
    return ok, s.x(), s.stats()


@pytest.mark.parametrize("P,L,fold", [(60, 600, 1), (257, 2600, 1), (257, 2600, 0), (1500, 15000, 1), (1500, 15000, 0), (4000, 40000, 1), (4000, 40000, 0),
                                      (13, 130, 1), (33, 300, 0)])
def test_band_kernel_matches_the_general_kernel_and_the_oracle(P, L, fold):
    pr = ba_case(P, L)
    lam = 25.0
    o = oracle_ba(pr)
    o.build_system()
    o.set_lambda(lam, True)
    assert o.solve()
    tol = dx_tolerance(o)[0] if P <= 300 else 1e-7
    ok1, x1, st1 = _solve(pr, lam, {"band_kernel": 1, "fuse_schur_reduce": fold})
    ok0, x0, st0 = _solve(pr, lam, {"band_kernel": 0, "fuse_schur_reduce": fold})
    assert ok0 and ok1
    assert st0["bandChains"] == 0
    if P >= 200:
        assert st1["bandChains"] > 0                      # (the kernel under test really ran)
    assert st1["choleskyNNZ"] == st0["choleskyNNZ"] and st1["numFronts"] == st0["numFronts"]
    assert relerr(x1, x0) < 1e-11
    assert relerr(x1, o.x()) < tol


def test_band_kernel_is_bit_repeatable_and_flags_a_non_positive_pivot():
    pr = ba_case(900, 9000)
    s = hip_ba(pr)
    s.buildSystem()
    xs = []
    for _ in range(3):
        s.setLambda(3.0, True)
        assert s.solve()
        s.restoreDiagonal()
        xs.append(s.x())
    assert s.stats()["bandChains"] > 0
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(xs[0], xs[2])
    s.setLambda(-1e9, True)                                # negative damping: not positive definite (block_solver.hpp:563-604 allows it)
    assert not s.solve()
    s.restoreDiagonal()
    s.setLambda(3.0, True)
    assert s.solve()
    s.restoreDiagonal()
    assert np.array_equal(s.x(), xs[0])


def test_band_kernel_with_huber_weights_in_an_lm_run():
    from openslam_g2o_amd import lm
    pr = ba_case(500, 5000, outlier_frac=0.05)
    out = []
    for band in (1, 0):
        s, g = lm.setup_device_ba(pr, huber_delta=1.0, options={"band_kernel": band})
        g.compute_active_errors()
        n, chis, lams, trials = lm.optimize(g, s, 5, "lm")
        out.append((n, chis, lams, trials, s.stats()["bandChains"]))
    assert out[0][4] > 0 and out[1][4] == 0
    assert out[0][0] == out[1][0] and out[0][3] == out[1][3]
    assert np.allclose(out[0][1], out[1][1], rtol=1e-9, atol=0) and np.allclose(out[0][2], out[1][2], rtol=1e-9, atol=0)


def test_graph_with_loop_closures_uses_the_general_kernels():
    from openslam_g2o_amd import synthetic as S
    pr = S.make_ba_loops(300, 1200, laps=3, hubs=1)
    Jp, Jc, err = S.ba_linearize(pr)
    pr.update(Jp=Jp, Jc=Jc, err=err, omega=S.ba_omega(pr))
    ok, x, st = _solve(pr, 30.0, {})
    o = oracle_ba(pr)
    o.build_system()
    o.set_lambda(30.0, True)
    assert ok and o.solve()
    assert relerr(x, o.x()) < dx_tolerance(o)[0]


@pytest.mark.parametrize("P,L,huber", [(257, 2600, 0.0), (1500, 15000, 0.0), (4000, 40000, 0.0), (9000, 45000, 1.0)])
def test_tree_levels_swept_by_groups_of_fronts_equal_the_task_by_task_sweep_bit_for_bit(P, L, huber):
    """Backward sweep of the tree levels by groups of fronts in one workgroup (tree_backward_kernel, option tree_backward) against
    the per-task sweep with a hand-off between workgroups per level: the same partial sums in the same order (bw_parts), so the
    solution is identical bit for bit; against the oracle to the usual tolerance."""
    pr = ba_case(P, L, outlier_frac=0.05 if huber else 0.0)
    lam = 25.0
    ok1, x1, st1 = _solve(pr, lam, {"tree_backward": 1}, huber=huber)     # tree levels by groups, leaf chains task by task (the default)
    ok2, x2, st2 = _solve(pr, lam, {"tree_backward": 2}, huber=huber)     # ... leaf chains one wave each
    ok0, x0, st0 = _solve(pr, lam, {"tree_backward": 0}, huber=huber)
    assert ok0 and ok1 and ok2
    assert st1["numFronts"] == st0["numFronts"]
    if P >= 1500:
        assert st1["treeBackwardGroups"] > 0 and st2["treeBackwardGroups"] > 0 and st0["treeBackwardGroups"] == 0     # (the kernels under test really ran)
    assert np.array_equal(x1, x0) and np.array_equal(x2, x0)
    o = oracle_ba(pr, huber=huber)
    o.build_system()
    o.set_lambda(lam, True)
    assert o.solve()
    assert relerr(x1, o.x()) < (dx_tolerance(o)[0] if P <= 300 else 1e-7)
