"""GPU: the device-resident BA front end (error / Jacobian producers, oplus, estimate stack) and a
full Levenberg-Marquardt run (config 5 semantics: Huber kernel, per-iteration lambda damping) against
the same loop driven through the CPU oracle -- chi2 trajectory, lambda sequence and LM trial counts."""
import numpy as np
import pytest

from openslam_g2o_amd import lm, synthetic as S
from oracle import oracle as O
from tests.helpers import ba_case, oracle_ba, relerr

pytestmark = pytest.mark.gpu


class OracleBAGraph:
    """Same protocol as lm.DeviceBAGraph, on the CPU oracle (types + solver)."""

    def __init__(self, prob, huber=0.0):
        self.pr = dict(prob)
        self.huber = huber
        self.o = O.OracleSolver(6, 3, prob["nP"], prob["nL"], True)
        self.k = self.o.add_edge_set(2, prob["v0"], prob["v1"])
        self.o.set_dims(self.k, 3, 6)
        self.o.build_structure()
        self.omega = S.ba_omega(prob)
        self.stack = []
        self.lm_hidx = np.arange(prob["L"], dtype=np.int32)

    def _edges(self, jac):
        p = self.pr
        return O.ba_edges(p["cams"], p["pts"], p["cam_idx"], p["pt_idx"], p["meas"], p["f"], p["cx"], p["cy"], jac=jac)

    def linearize(self):
        Jp, Jc, err = self._edges(True)
        self.o.set_edge_data(self.k, Jp, Jc, self.omega, err, self.huber)
        self._J = (Jp, Jc)

    def compute_active_errors(self):
        err = self._edges(False)
        self.o.set_edge_data(self.k, self._J[0], self._J[1], self.omega, err, self.huber)

    def chi2(self):
        return self.o.chi2()

    def update(self):
        p = self.pr
        cams, pts = O.ba_oplus(p["cams"], p["pts"], p["cam_hidx"], self.lm_hidx, self.o.x(), 6 * p["nP"])
        p["cams"], p["pts"] = cams, pts

    def push(self):
        self.stack.append((self.pr["cams"].copy(), self.pr["pts"].copy()))

    def pop(self):
        self.pr["cams"], self.pr["pts"] = self.stack.pop()

    def discard_top(self):
        self.stack.pop()


class OracleSolverAdapter:
    def __init__(self, o):
        self.o = o

    def buildSystem(self):
        self.o.build_system()

    def setLambda(self, lam, backup=False):
        self.o.set_lambda(lam, backup)

    def restoreDiagonal(self):
        self.o.restore_diagonal()

    def solve(self):
        return self.o.solve()

    def maxDiagonal(self):
        return self.o.max_diagonal()

    def computeScale(self, lam):
        return self.o.compute_scale(lam)

    def b(self):
        return self.o.b()

    def x(self):
        return self.o.x()

    def setX(self, v):
        self.o.view("x", self.o.n)[:] = v

    def multiplyHessian(self, v):
        return self.o.multiply_full(v)


@pytest.mark.parametrize("fused", [1, 0])
def test_device_producers_match_oracle(fused):
    """fused=1: errors/Jacobians evaluated inside the assembly kernels; fused=0: separate linearize
    kernel filling the Jacobian arrays of the generic edge-data path."""
    pr = ba_case(30, 300)
    s, g = lm.setup_device_ba(pr)
    s.setOption("ba_fused", fused)
    g.linearize()
    s.buildSystem()
    o = oracle_ba(pr)
    o.build_system()
    assert abs(s.chi2() - o.chi2()) <= 1e-12 * o.chi2()
    assert relerr(s.b(), o.b()) < 1e-12                       # device Jacobians == host Jacobians
    # oplus: apply a solver step on both sides and compare the states
    s.setLambda(10.0, True)
    o.set_lambda(10.0, True)
    assert s.solve() and o.solve()
    g.push()
    g.update()
    cams, pts = s.baGetEstimates()
    cams_o, pts_o = O.ba_oplus(pr["cams"], pr["pts"], pr["cam_hidx"], np.arange(pr["L"], dtype=np.int32), o.x(), 6 * pr["nP"])
    assert relerr(cams, cams_o) < 1e-10 and relerr(pts, pts_o) < 1e-10
    assert np.array_equal(cams[:2], pr["cams"][:2])            # fixed poses untouched
    # selected vertices only (g2ohip_ba_get_estimates_of: what a caller with a few host-side edges reads of a trial)
    ci, pi = [3, 0, pr["P"] - 1], [7, pr["L"] - 1, 7, 0]
    cs, ps = s.baGetEstimatesOf(ci, pi)
    assert np.array_equal(cs, cams[ci]) and np.array_equal(ps, pts[pi])
    cs, ps = s.baGetEstimatesOf([], [5])
    assert cs.shape == (0, 12) and np.array_equal(ps, pts[[5]])
    with pytest.raises(Exception):
        s.baGetEstimatesOf([pr["P"]], [])                    # out of range
    g.pop()
    cams2, pts2 = s.baGetEstimates()
    assert np.array_equal(cams2, pr["cams"]) and np.array_equal(pts2, pr["pts"])   # pop restores bit-exactly


@pytest.mark.parametrize("K", [2, 11, 23])
def test_fused_assembly_with_short_and_long_observation_lists(K):
    """Landmarks with fewer / more observations than the 8 lanes of their group: the looped lanes, the LDS-staged
    Hpl stream-out (a wave's blocks beyond its 64 slots fall back to direct stores) and the pose-major copies."""
    pr = ba_case(40, 90, obs_per_landmark=K)
    s, g = lm.setup_device_ba(pr)
    g.linearize()
    s.buildSystem()
    o = oracle_ba(pr)
    o.build_system()
    from openslam_g2o_amd import capi
    for which, name in ((capi.HPP, "Hpp"), (capi.HPL, "Hpl"), (capi.HLL, "Hll")):
        assert relerr(s.values(which), o.values(name)) < 1e-12, name
    assert relerr(s.b(), o.b()) < 1e-12
    s.setLambda(5.0, True)
    o.set_lambda(5.0, True)
    assert s.solve() and o.solve()
    assert relerr(s.x(), o.x()) < 1e-8


@pytest.mark.parametrize("P,L,laps,hubs,stride", [(150, 700, 3, 0, 3), (320, 1500, 5, 0, 3), (320, 1500, 5, 3, 3), (640, 2500, 8, 2, 3),
                                                    (2400, 12000, 6, 3, 24)])
def test_ba_graph_with_loop_closures_and_ragged_lists(P, L, laps, hubs, stride):
    """Not a band: the camera passes the same places several times (synthetic.make_ba_loops), so the reduced system
    couples distant poses (frontal matrices of a few hundred rows: the LDS and scratch-slab kernels under the virtual
    Schur source), observation lists are ragged (2 .. 5 laps) and, with hubs, a few points are seen by more poses than
    a wavefront has lanes (the tiles then leave the landmark side to the stand-alone kernel).  System, solution and one
    LM step against the oracle."""
    from openslam_g2o_amd import capi
    pr = S.make_ba_loops(P, L, laps=laps, hubs=hubs, hub_stride=stride)   # (the last case: 2 398 free poses, hubs seen by 100 poses each)
    Jp, Jc, err = S.ba_linearize(pr)
    pr.update(Jp=Jp, Jc=Jc, err=err, omega=S.ba_omega(pr))
    o = oracle_ba(pr, huber=2.0)
    o.build_system()
    s, g = lm.setup_device_ba(pr, huber_delta=2.0)
    g.linearize()
    assert abs(g.chi2() - o.chi2()) <= 1e-10 * o.chi2()
    s.buildSystem()
    lam = 1e-4 * o.max_diagonal()
    s.setLambda(lam, True)
    o.set_lambda(lam, True)
    assert s.solve() and o.solve()
    assert relerr(s.x(), o.x()) < 1e-7
    x = s.x()
    assert s.solve() and np.array_equal(s.x(), x)        # bit-repeatable
    assert relerr(s.values(capi.HSCHUR), o.values("Hschur")) < 1e-11   # (the damped reduced system of this solve)
    s.restoreDiagonal()
    o.restore_diagonal()
    for which, name in ((capi.HPP, "Hpp"), (capi.HPL, "Hpl"), (capi.HLL, "Hll")):
        assert relerr(s.values(which), o.values(name)) < 1e-11, name
    assert relerr(s.b(), o.b()) < 1e-12
    st = s.stats()
    assert st["maxFrontDim"] > 72                           # (the band case stays at 72)
    # one LM step on the device
    chi0 = g.chi2()
    s.setLambda(lam, True)
    assert s.solve()
    g.push(); g.update(); s.restoreDiagonal(); g.compute_active_errors()
    assert g.chi2() < chi0


@pytest.mark.parametrize("K", [1, 2, 5, 11, 23, 70])
def test_schur_tiles_assemble_their_landmarks(K):
    """Default fused path: build_system launches nothing for the landmark side, the Schur tiles of solve() produce Hll, b_l
    and the errors (one lane per observation, per-tile slot tables; lists of 1 .. 64 observations, longer ones fall back to
    the stand-alone kernel).  Nothing is read between build_system and solve here, so the tile variant is what runs;
    afterwards Hll / b / Dinv / x are compared with the oracle and with the stand-alone path (ba_fuse_landmarks = 0)."""
    from openslam_g2o_amd import capi
    pr = ba_case(80, 60, obs_per_landmark=K) if K >= 70 else ba_case(40, 90, obs_per_landmark=K)
    o = oracle_ba(pr, huber=1.5)
    o.build_system()
    o.set_lambda(5.0, True)
    assert o.solve()
    got = {}
    for fuse in (1, 0):
        s, g = lm.setup_device_ba(pr, huber_delta=1.5)
        s.setOption("ba_fuse_landmarks", fuse)
        g.linearize()
        s.buildSystem()
        s.setLambda(5.0, True)
        assert s.solve()
        x = s.x()
        s.restoreDiagonal()
        got[fuse] = (x, s.b(), s.values(capi.HLL), s.values(capi.DINV))
    o.restore_diagonal()
    for fuse in (1, 0):
        x, b, hll, dinv = got[fuse]
        assert relerr(x, o.x()) < 1e-8, fuse
        assert relerr(b, o.b()) < 1e-12, fuse
        assert relerr(hll, o.values("Hll")) < 1e-12, fuse
    # the two paths differ by the summation order of a landmark's observations only
    assert relerr(got[1][0], got[0][0]) < 1e-8   # (both within 1e-8 of the oracle; lists of one observation condition the system badly)
    assert relerr(got[1][2], got[0][2]) < 1e-13 and relerr(got[1][1], got[0][1]) < 1e-13


@pytest.mark.parametrize("huber,outliers", [(0.0, 0.0), (1.0, 0.05)])
def test_lm_trajectory_matches_oracle(huber, outliers):
    pr = ba_case(60, 600, outlier_frac=outliers)
    s, g = lm.setup_device_ba(pr, huber_delta=huber)
    n_gpu, chi_gpu, lam_gpu, tr_gpu = lm.optimize(g, s, 8, "lm")
    og = OracleBAGraph(pr, huber)
    n_cpu, chi_cpu, lam_cpu, tr_cpu = lm.optimize(og, OracleSolverAdapter(og.o), 8, "lm")
    assert n_gpu == n_cpu and tr_gpu == tr_cpu                 # same accept/reject decisions
    assert np.allclose(chi_gpu, chi_cpu, rtol=1e-6, atol=0)    # north_star bar: chi2 within 1e-6 relative
    assert np.allclose(lam_gpu, lam_cpu, rtol=1e-6, atol=0)
    assert chi_gpu[-1] <= chi_gpu[0]
    if huber == 0.0:
        assert chi_gpu[-1] < 3.0 * pr["E"]                      # converged to the pixel-noise level (sigma = 1)
    cams, pts = s.baGetEstimates()
    assert relerr(cams, og.pr["cams"]) < 1e-6 and relerr(pts, og.pr["pts"]) < 1e-6


def test_lm_trajectory_on_a_graph_with_loop_closures_matches_oracle():
    """Levenberg-Marquardt with Huber kernels on the graph that is not a band (loop closures, ragged lists, a hub point
    seen by more poses than a Schur tile holds): accept / reject decisions, chi2 and lambda against the oracle-driven
    loop."""
    pr = S.make_ba_loops(260, 1100, laps=4, hubs=1)
    s, g = lm.setup_device_ba(pr, huber_delta=1.5)
    n_gpu, chi_gpu, lam_gpu, tr_gpu = lm.optimize(g, s, 6, "lm")
    og = OracleBAGraph(pr, 1.5)
    n_cpu, chi_cpu, lam_cpu, tr_cpu = lm.optimize(og, OracleSolverAdapter(og.o), 6, "lm")
    assert n_gpu == n_cpu and tr_gpu == tr_cpu
    assert np.allclose(chi_gpu, chi_cpu, rtol=1e-6, atol=0)
    assert np.allclose(lam_gpu, lam_cpu, rtol=1e-6, atol=0)
    assert chi_gpu[-1] < chi_gpu[0]


def test_dogleg_trajectory_matches_oracle():
    """OptimizationAlgorithmDogleg restated in lm.py (optimization_algorithm_dogleg.cpp:57-207) over the device-resident
    BA graph against the same driver over the CPU oracle: chi2 trajectory, trust-region radius and step types."""
    pr = ba_case(40, 400)
    s, g = lm.setup_device_ba(pr)
    n_it = 6
    done, chis, deltas, trials = lm.optimize(g, s, n_it, algorithm="dogleg")
    og = OracleBAGraph(pr)
    done_o, chis_o, deltas_o, trials_o = lm.optimize(og, OracleSolverAdapter(og.o), n_it, algorithm="dogleg")
    # trial counts are compared while the oracle's chi2 still decreases by more than the rounding level of the
    # sum (1e-10 relative): at the converged point the sign of the "gain" is decided by the last bits of dx, and a
    # rejected step there only shrinks the trust region (optimization_algorithm_dogleg.cpp:166-199)
    assert done == done_o
    sig = [k for k in range(len(trials_o)) if k == 0 or (chis_o[k - 1] - chis_o[k]) > 1e-10 * chis_o[k - 1]]
    assert len(sig) >= 4 and [trials[k] for k in sig] == [trials_o[k] for k in sig]
    assert relerr(chis, chis_o) < 1e-7 and relerr([deltas[k] for k in sig], [deltas_o[k] for k in sig]) < 1e-6
    assert all(b <= a for a, b in zip(chis, chis[1:])) and chis[-1] < chis[0]
    # a small trust region forces the steepest-descent and dogleg branches
    s2, g2 = lm.setup_device_ba(pr)
    og2 = OracleBAGraph(pr)
    r1 = lm.optimize(g2, s2, 5, algorithm="dogleg", initial_delta=0.05)
    r2 = lm.optimize(og2, OracleSolverAdapter(og2.o), 5, algorithm="dogleg", initial_delta=0.05)
    assert r1[0] == r2[0] and r1[3] == r2[3] and relerr(r1[1], r2[1]) < 1e-7 and relerr(r1[2], r2[2]) < 1e-6


def test_trial_stats_matches_the_three_separate_calls():
    """g2ohip_solve_async + g2ohip_trial_stats (one synchronisation per LM trial) against solve / chi2 / computeScale, and
    the failure flag of an indefinite system arriving through the deferred status."""
    pr = ba_case(50, 500)
    a, ga = lm.setup_device_ba(pr)
    b, gb = lm.setup_device_ba(pr)
    for s, g in ((a, ga), (b, gb)):
        g.linearize()
        s.buildSystem()
        s.setLambda(3.0, True)
    assert a.solve()
    ga.update(); a.restoreDiagonal(); ga.compute_active_errors()
    chi, sc = a.chi2(), a.computeScale(3.0)
    b.solveAsync()
    gb.update(); b.restoreDiagonal(); gb.compute_active_errors()
    ok, chi_b, sc_b = b.trialStats(3.0)
    assert ok and chi_b == chi and sc_b == sc
    assert np.array_equal(a.x(), b.x())
    # without a pending solve the call only evaluates the sums (chi2 comes out of the cache)
    ok, chi_c, sc_c = b.trialStats(3.0)
    assert ok and chi_c == chi and sc_c == sc
    b.buildSystem()
    b.setLambda(-1e3 * b.maxDiagonal(), True)
    b.solveAsync()
    ok, _, _ = b.trialStats(1.0)
    assert not ok
    b.restoreDiagonal()
    b.setLambda(3.0, True)
    assert b.solve()
    # the two halves: trialStatsBegin queues the sums and their read-back (no synchronisation), the next trialStats only waits
    # for them -- other work queued in between (here: the estimates pushed back and forth) does not disturb the numbers
    c, gc = lm.setup_device_ba(pr)
    gc.linearize()
    c.buildSystem()
    c.setLambda(3.0, True)
    c.solveAsync()
    gc.update(); c.restoreDiagonal(); gc.compute_active_errors()
    c.trialStatsBegin(3.0)
    with pytest.raises(Exception):
        c.trialStatsBegin(3.0)        # the previous one has not been read
    ok, chi_d, sc_d = c.trialStats(123.0)   # (lambda of the begun call counts)
    assert ok and chi_d == chi and sc_d == sc
    assert np.array_equal(a.x(), c.x())


def test_batch_statistics_line_is_g2o_stats_compatible():
    """G2OBatchStatistics (batch_stats.h:40-77) through lm.optimize(stats=...): every field of the `g2o -stats` line, in the
    reference's order and "name= value<TAB> " format (batch_stats.cpp:49-82); the solver-side fields come from
    g2ohip_get_stats (incl. the device front end's timeResiduals / timeLinearize / timeUpdate and the dependency-launch
    fallback counter, expected 0)."""
    import re
    pr = ba_case(60, 600)
    s, g = lm.setup_device_ba(pr)
    s.setProfiling(True)
    stats = []
    n, chis, lams, trials = lm.optimize(g, s, 3, "lm", stats=stats, num_vertices=pr["nP"] + pr["nL"], num_edges=pr["E"])
    assert len(stats) == n == 3
    for it, d in enumerate(stats):
        line = lm.format_batch_stats(d)
        names = re.findall(r"(\w+)= ", line)
        assert names == list(lm.BATCH_STAT_FIELDS) and line.count("\t ") == len(names)
        assert d["iteration"] == it and d["levenbergIterations"] == trials[it] and d["chi2"] == chis[it]
        assert d["hessianPoseDimension"] == 6 * pr["nP"] and d["hessianLandmarkDimension"] == 3 * pr["nL"]
        # (the stage timers of the linear solve are filled by solve(); the device-resident trial queues the whole solve
        # asynchronously -- g2ohip_solve_async -- and leaves them at the last synchronous value)
        assert d["choleskyNNZ"] > 0 and d["timeIteration"] > 0 and d["timeNumericDecomposition"] >= 0
        assert d["timeSchurComplement"] >= 0 and d["timeUpdate"] > 0 and d["timeResiduals"] > 0
        assert d["dependencyFallbacks"] == 0



def test_front_end_rejects_inconsistent_indices():
    """g2ohip_ba_set_edges / g2ohip_ba_set_estimates validate the edge -> estimate indices and their hessian indices
    against the edge set (an out-of-range index would be an out-of-bounds device read, a mismatch a silently inconsistent
    system); the wrappers check array lengths."""
    from openslam_g2o_amd import capi
    pr = ba_case(20, 120)
    s = capi.HipBlockSolver(6, 3, 0)
    k = s.addEdgeSet(2, pr["v0"], pr["v1"])
    s.buildStructure(pr["nP"], pr["nL"], True)
    with pytest.raises(ValueError):
        s.baSetEdges(k, pr["cam_idx"][:-1], pr["pt_idx"], pr["meas"], None, pr["f"], pr["cx"], pr["cy"])
    s.baSetEdges(k, pr["cam_idx"], pr["pt_idx"], pr["meas"], None, pr["f"], pr["cx"], pr["cy"])
    bad = pr["cam_hidx"].copy()
    bad[5], bad[6] = bad[6], bad[5]                      # two cameras swap their hessian indices
    with pytest.raises(capi.G2oHipError):
        s.baSetEstimates(pr["cams"], bad, pr["pts"], np.arange(pr["L"], dtype=np.int32))
    with pytest.raises(capi.G2oHipError):                # a point table that is too short
        s.baSetEstimates(pr["cams"], pr["cam_hidx"], pr["pts"][:-3], np.arange(pr["L"] - 3, dtype=np.int32))
    s.baSetEstimates(pr["cams"], pr["cam_hidx"], pr["pts"], np.arange(pr["L"], dtype=np.int32))
    s.baLinearize(True)
    s.buildSystem()
    s.setLambda(1.0, True)
    assert s.solve()
