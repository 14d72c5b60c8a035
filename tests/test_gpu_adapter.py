"""The g2o adapter EXECUTED: openslam_g2o_amd/cpp/solver_hip.cpp + g2o_hip_solver.h are compiled against the test host
tests/cpp/mini_g2o (this repository's own small implementation of the g2o interfaces the adapter touches -- Eigen and g2o
cannot be installed here), linked with libg2ohip.so into libg2o_solver_hip.so, loaded with dlopen by a host program and
found BY NAME through the optimisation-algorithm factory, the way the g2o CLI finds a solver plugin
(g2o_common.cpp:81-167, optimization_algorithm_factory.h:120-162).  Levenberg-Marquardt then runs through the
g2o::OptimizationAlgorithm -> g2o::Solver (wide seam, generic path and device fast path) and g2o::BlockSolver ->
g2o::LinearSolver (narrow seam) vtables on the GPU; chi2, lambda, trial counts and the final estimates are compared with the
same loop driven through the CPU oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from openslam_g2o_amd import lm, synthetic as S
from tests.helpers import ba_case, oracle_ba, relerr

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "tests", "cpp", "mini_g2o")
BUILD = os.path.join(HOST_DIR, "build")


@pytest.fixture(scope="module")
def host():
    subprocess.run(["make", "-s", "-C", HOST_DIR], check=True, capture_output=True, text=True)
    return os.path.join(BUILD, "g2o_host"), os.path.join(BUILD, "libg2o_solver_hip.so")


def _write_problem(path, pr, huber=0.0, fix_points=False):
    with open(path, "w") as f:
        f.write("%d %d %d %.17g %.17g %.17g %.17g\n" % (pr["P"], pr["L"], pr["E"], pr["f"], pr["cx"], pr["cy"], huber))
        for i in range(pr["P"]):
            f.write("%d %s\n" % (1 if pr["cam_hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in pr["cams"][i])))
        for j in range(pr["L"]):
            f.write("%d %s\n" % (1 if fix_points else 0, " ".join("%.17g" % v for v in pr["pts"][j])))
        for k in range(pr["E"]):
            f.write("%d %d %.17g %.17g\n" % (pr["cam_idx"][k], pr["pt_idx"][k], pr["meas"][k][0], pr["meas"][k][1]))


def _run(host, problem, solver, iterations, out, env=None, mode=None):
    exe, plugin = host
    e = dict(os.environ)
    e["G2OHIP_ADAPTER_VERBOSE"] = "1"
    e.update(env or {})
    cmd = [exe, problem, plugin, solver, str(iterations), out] + ([mode] if mode else [])
    r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    return json.load(open(out)), r.stderr


def _oracle_lm(pr, iterations, huber=0.0):
    from tests.test_gpu_lm import OracleBAGraph, OracleSolverAdapter
    g = OracleBAGraph(pr, huber=huber)
    g.compute_active_errors = g.compute_active_errors   # (protocol object)
    g.linearize()
    chi0 = g.chi2()
    n, chis, lams, trials = lm.optimize(g, OracleSolverAdapter(g.o), iterations, "lm")
    return chi0, n, chis, lams, trials, g.pr["cams"], g.pr["pts"]


@pytest.mark.parametrize("huber", [0.0, 1.0])
def test_levenberg_marquardt_through_the_g2o_vtables(host, tmp_path, huber):
    pr = ba_case(40, 400, outlier_frac=0.05 if huber > 0 else 0.0)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr, huber)
    chi0, n_o, chis_o, lams_o, trials_o, cams_o, pts_o = _oracle_lm(pr, 5, huber)
    runs = {}
    for tag, solver, env in (("wide-fast", "lm_fix6_3_hip", {}), ("wide-generic", "lm_fix6_3_hip", {"G2OHIP_ADAPTER_FASTPATH": "0"}),
                             ("narrow", "lm_fix6_3_hipls", {})):
        out, err = _run(host, prob, solver, 5, str(tmp_path / (tag + ".json")), env)
        runs[tag] = out
        # registered through the factory with the property table of solver_csparse.cpp:117-140
        assert out["property"] == {"name": solver, "type": "MI355X HIP", "requiresMarginalize": True, "poseDim": 6, "landmarkDim": 3}
        assert ("device front end (g2ohip_ba_*) for %d EdgeProjectXYZ2UV" % pr["E"] in err) == (tag == "wide-fast"), err[-500:]
        assert abs(out["chi2_initial"] - chi0) <= 1e-9 * chi0
        assert out["iterations"] == n_o and out["trials"] == trials_o, (tag, out["trials"], trials_o)
        assert np.allclose(out["chi2"], chis_o, rtol=1e-7, atol=0), (tag, out["chi2"], chis_o)
        assert np.allclose(out["lambda"], lams_o, rtol=1e-7, atol=0), (tag, out["lambda"], lams_o)
        assert relerr(np.array(out["cams"]).reshape(-1, 12), cams_o) < 1e-7 and relerr(np.array(out["points"]).reshape(-1, 3), pts_o) < 1e-7
    assert runs["wide-fast"]["chi2"][-1] < 0.6 * chi0
    # the device front end and the uploaded Jacobians walk the same trajectory
    assert np.allclose(runs["wide-fast"]["chi2"], runs["wide-generic"]["chi2"], rtol=1e-9, atol=0)


def test_two_camera_parameters_and_a_robust_kernel_on_half_the_edges_stay_on_the_device(host, tmp_path):
    """Every EdgeProjectXYZ2UV carries its own CameraParameters (types_six_dof_expmap.h:133-153) and its own robust kernel: a
    graph with two intrinsics and Huber on every other edge is FOUR (intrinsics, kernel) combinations.  The adapter binds
    them to the device front end as one edge set with four edge classes (g2ohip_ba_set_edges_classes) instead of leaving
    three of four groups on the generic path; both paths and the oracle-driven loop walk the same trajectory."""
    from tests.test_gpu_edge_classes import CLASSES, OracleClassBAGraph
    from tests.test_gpu_lm import OracleSolverAdapter
    pr = ba_case(40, 400, outlier_frac=0.05)
    second = (pr["cam_idx"] % 2) == 1
    robust = (np.arange(pr["E"]) % 2) == 1
    # classes of tests/test_gpu_edge_classes.CLASSES: 0 / 1 first camera plain / Huber(1.0), 2 / 3 second camera plain / Huber(1.5)
    cls = (2 * second + robust).astype(np.int32)
    f0, c0 = pr["f"], np.array([pr["cx"], pr["cy"]])
    pr["meas"] = (pr["meas"] - c0) / f0 * CLASSES[cls, 0][:, None] + CLASSES[cls, 1:3]
    pr["edge_class"] = cls
    prob = str(tmp_path / "p.txt")
    with open(prob, "w") as f:
        f.write("%d %d %d %.17g %.17g %.17g 0 %.17g %.17g %.17g\n" % (pr["P"], pr["L"], pr["E"], *CLASSES[0, :3], *CLASSES[2, :3]))
        for i in range(pr["P"]):
            f.write("%d %s\n" % (1 if pr["cam_hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in pr["cams"][i])))
        for j in range(pr["L"]):
            f.write("0 %s\n" % " ".join("%.17g" % v for v in pr["pts"][j]))
        for k in range(pr["E"]):
            f.write("%d %d %.17g %.17g %d %.17g\n" % (pr["cam_idx"][k], pr["pt_idx"][k], pr["meas"][k][0], pr["meas"][k][1], int(second[k]),
                                                   CLASSES[cls[k], 4]))
    go = OracleClassBAGraph(pr)
    go.linearize()
    chi0 = go.chi2()
    n_o, chis_o, lams_o, trials_o = lm.optimize(go, OracleSolverAdapter(go.o), 5, "lm")
    runs = {}
    for tag, env in (("fast", {}), ("generic", {"G2OHIP_ADAPTER_FASTPATH": "0"})):
        out, err = _run(host, prob, "lm_fix6_3_hip", 5, str(tmp_path / (tag + ".json")), env, mode="classes")
        runs[tag] = out
        assert ("device front end (g2ohip_ba_*) for %d EdgeProjectXYZ2UV" % pr["E"] in err and "4 edge classes" in err) == (tag == "fast"), err[-600:]
        assert abs(out["chi2_initial"] - chi0) <= 1e-9 * chi0
        assert out["iterations"] == n_o and out["trials"] == trials_o, (tag, out["trials"], trials_o)
        assert np.allclose(out["chi2"], chis_o, rtol=1e-7, atol=0), (tag, out["chi2"], chis_o)
        assert np.allclose(out["lambda"], lams_o, rtol=1e-7, atol=0)
        assert relerr(np.array(out["cams"]).reshape(-1, 12), go.pr["cams"]) < 1e-7
    assert np.allclose(runs["fast"]["chi2"], runs["generic"]["chi2"], rtol=1e-9, atol=0)


def test_pose_graph_with_kernels_on_its_loop_closures_binds_as_one_group(host, tmp_path):
    """manhattan3500 with a Huber kernel on the loop closures only, through the vtables: ONE group of EdgeSE2 whose edges differ
    in their robust kernel -- bound to the pose-graph front end with one kernel per edge (g2ohip_set_robust_kernel_per_edge); the
    fast path and the generic path (the same per-edge kernels over uploaded Jacobians) walk one trajectory, which differs from
    the plain graph's."""
    from tests.helpers import manhattan_golden
    g = manhattan_golden()
    path = str(tmp_path / "m.txt")
    with open(path, "w") as f:
        nv, ne = len(g["estimates"]), len(g["vi"])
        f.write("%d %d\n" % (nv, ne))
        for i in range(nv):
            f.write("%d %s\n" % (1 if g["hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in g["estimates"][i])))
        for k in range(ne):
            f.write("%d %d %s %s\n" % (g["vi"][k], g["vj"][k], " ".join("%.17g" % v for v in g["meas"][k]),
                                       " ".join("%.17g" % v for v in np.asarray(g["omega"][k]).reshape(-1))))
    runs = []
    for env in ({}, {"G2OHIP_ADAPTER_FASTPATH": "0"}):
        out, err = _run(host, path, "lm_fix3_2_hip", 6, str(tmp_path / "o.json"), env, mode="se2huber:1.5")
        assert ("device fast path for %d EdgeSE2" % len(g["vi"]) in err) == (not env), err[-400:]
        assert out["iterations"] == 6 and all(b <= a * (1 + 1e-12) for a, b in zip([out["chi2_initial"]] + out["chi2"], out["chi2"]))
        runs.append(out)
    assert np.allclose(runs[0]["chi2"], runs[1]["chi2"], rtol=1e-9, atol=0)
    assert np.abs(np.array(runs[0]["poses"]) - np.array(runs[1]["poses"])).max() < 1e-7
    plain, _ = _run(host, path, "lm_fix3_2_hip", 6, str(tmp_path / "p.json"), {}, mode="se2")
    assert runs[0]["chi2_initial"] < plain["chi2_initial"] * (1 - 1e-3)      # (the kernels bite: robustified chi2 of the same graph)


def test_edge_classes_refused_by_the_front_end_fall_back_to_generic_groups(host, tmp_path):
    """Edge classes need the fused device path; a FIXED point takes a landmark off it (its edges would be skipped by the landmark-major
    kernels), so g2ohip_ba_set_edges_classes refuses the merged group.  The adapter then registers the edges again as generic groups
    (one per shape, one robust kernel per edge) -- same trajectory as with the fast path switched off, and the plugin says so."""
    from tests.test_gpu_edge_classes import CLASSES
    pr = ba_case(30, 200, outlier_frac=0.05)
    second = (pr["cam_idx"] % 2) == 1
    robust = (np.arange(pr["E"]) % 2) == 1
    cls = (2 * second + robust).astype(np.int32)
    pr["meas"] = (pr["meas"] - np.array([pr["cx"], pr["cy"]])) / pr["f"] * CLASSES[cls, 0][:, None] + CLASSES[cls, 1:3]
    prob = str(tmp_path / "p.txt")
    with open(prob, "w") as f:
        f.write("%d %d %d %.17g %.17g %.17g 0 %.17g %.17g %.17g\n" % (pr["P"], pr["L"], pr["E"], *CLASSES[0, :3], *CLASSES[2, :3]))
        for i in range(pr["P"]):
            f.write("%d %s\n" % (1 if pr["cam_hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in pr["cams"][i])))
        for j in range(pr["L"]):
            f.write("%d %s\n" % (1 if j == 3 else 0, " ".join("%.17g" % v for v in pr["pts"][j])))     # point 3 is fixed
        for k in range(pr["E"]):
            f.write("%d %d %.17g %.17g %d %.17g\n" % (pr["cam_idx"][k], pr["pt_idx"][k], pr["meas"][k][0], pr["meas"][k][1], int(second[k]),
                                                   CLASSES[cls[k], 4]))
    runs = {}
    for tag, env in (("fast", {}), ("generic", {"G2OHIP_ADAPTER_FASTPATH": "0"})):
        out, err = _run(host, prob, "lm_fix6_3_hip", 4, str(tmp_path / (tag + ".json")), env, mode="classes")
        runs[tag] = out
        assert "device front end (g2ohip_ba_*)" not in err
        assert ("fast path not available" in err) == (tag == "fast"), err[-600:]
        assert out["iterations"] == 4 and out["chi2"][-1] < out["chi2_initial"]
    assert np.allclose(runs["fast"]["chi2"], runs["generic"]["chi2"], rtol=1e-12, atol=0)
    assert runs["fast"]["trials"] == runs["generic"]["trials"]


def test_gauss_newton_and_dogleg_creators_construct(host, tmp_path):
    pr = ba_case(12, 100)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr)
    out, _ = _run(host, prob, "gn_fix6_3_hip", 3, str(tmp_path / "gn.json"))
    assert out["iterations"] == 3 and out["chi2"][-1] < out["chi2_initial"]
    out, _ = _run(host, prob, "gn_fix6_3_hipls", 2, str(tmp_path / "gnls.json"))
    assert out["iterations"] == 2 and out["chi2"][-1] < out["chi2_initial"]
    out, _ = _run(host, prob, "dl_fix6_3_hip", 0, str(tmp_path / "dl.json"))     # constructed via dynamic_cast<BlockSolverBase*>
    assert out["property"]["name"] == "dl_fix6_3_hip"


@pytest.mark.parametrize("solver", ["lm_fix6_3_hip", "lm_fix6_3_hipls"])
def test_compute_marginals_through_the_seams(host, tmp_path, solver):
    """Solver::computeMarginals: on the narrow seam g2o's BlockSolver calls solve(Hschur) and then solvePattern(Hpp) on the
    SAME LinearSolver without init() (block_solver.hpp:489-499) -- a second pattern between two init() calls."""
    pr = ba_case(30, 300)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr)
    out, _ = _run(host, prob, solver, 0, str(tmp_path / "m.json"), mode="marginals")
    assert out["solve_ok"] and out["marginals_ok"]
    o = oracle_ba(pr)
    o.build_system()
    assert o.solve()
    assert relerr(np.array(out["x"]), o.x()) < 1e-7
    nP = pr["nP"]
    cp, ri = o.pattern("pp")
    V = o.values("Hpp").reshape(-1, 6, 6)
    H = np.zeros((6 * nP, 6 * nP))
    for c in range(nP):
        for q in range(cp[c], cp[c + 1]):
            r = ri[q]
            H[6 * r:6 * r + 6, 6 * c:6 * c + 6] = V[q].T
            H[6 * c:6 * c + 6, 6 * r:6 * r + 6] = V[q]
    Hi = np.linalg.inv(H)
    assert len(out["blocks"]) >= 6
    for b in out["blocks"]:
        got = np.array(b["v"]).reshape(6, 6).T          # column-major
        ref = Hi[6 * b["r"]:6 * b["r"] + 6, 6 * b["c"]:6 * b["c"] + 6]
        assert np.abs(got - ref).max() <= 1e-8 * np.abs(Hi).max(), (b["r"], b["c"])


def test_localisation_graph_with_every_point_fixed(host, tmp_path):
    """All vertices of one side of a group fixed (EdgeProjectXYZ2UV over fixed points): the generic path's Jacobian buffers
    are sized for what the library reads (round-2 advisor finding), the fast path declines (no marginalized point)."""
    pr = ba_case(10, 120)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr, fix_points=True)
    for env in ({}, {"G2OHIP_ADAPTER_FASTPATH": "0"}):
        out, _ = _run(host, prob, "lm_fix6_3_hip", 3, str(tmp_path / "loc.json"), env)
        assert out["iterations"] >= 1 and out["chi2"][-1] < out["chi2_initial"]
        assert np.abs(np.array(out["points"]).reshape(-1, 3) - pr["pts"]).max() == 0.0      # fixed points did not move


def test_online_growth_through_update_structure(host, tmp_path):
    """Solver::updateStructure through the vtables (block_solver.hpp:297-351; SparseOptimizer::updateInitialization,
    sparse_optimizer.cpp:445-479): a localisation graph (every point fixed: no Schur complement) is optimised for its first
    cameras, grown by the others, and optimised on.  With fixed points the cameras are independent of each other, so every
    camera must end where the same number of Gauss-Newton steps takes it in a run over the whole graph."""
    pr = ba_case(12, 150)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr, fix_points=True)
    n_first, it = 7, 2
    grown, err = _run(host, prob, "gn_fix6_3_hip", it, str(tmp_path / "on.json"), mode="online:%d" % n_first)
    assert grown["iterations"] == 2 * it
    short, _ = _run(host, prob, "gn_fix6_3_hip", it, str(tmp_path / "a.json"))
    long_, _ = _run(host, prob, "gn_fix6_3_hip", 2 * it, str(tmp_path / "b.json"))
    cg, cs, cl = (np.array(o["cams"]).reshape(-1, 12) for o in (grown, short, long_))
    free = pr["cam_hidx"] >= 0
    first = np.arange(pr["P"]) < n_first
    assert np.abs(cg[first & free] - cl[first & free]).max() < 1e-9      # 2 + 2 steps
    assert np.abs(cg[~first & free] - cs[~first & free]).max() < 1e-9    # added later: 2 steps
    assert np.abs(cg[~free] - pr["cams"][~free]).max() == 0.0            # fixed cameras did not move
    assert grown["chi2"][-1] <= grown["chi2"][it] * (1 + 1e-9)            # (chi2_initial counts the first graph's edges only)
    # a graph with marginalised points: the growth is refused (the reference aborts), the host reports it
    prob2 = str(tmp_path / "q.txt")
    _write_problem(prob2, pr)
    exe, plugin = host
    r = subprocess.run([exe, prob2, plugin, "gn_fix6_3_hip", "1", str(tmp_path / "c.json"), "online:%d" % n_first], capture_output=True, text=True, timeout=600)
    assert r.returncode == 5 and "Schur not supported" in r.stderr


@pytest.mark.parametrize("solver", ["lm_fix6_3_hip", "lm_fix6_3_hipls"])
def test_a_second_optimize_on_the_same_solver(host, tmp_path, solver):
    """optimize() twice on one optimizer: buildStructure runs again at the second iteration 0
    (optimization_algorithm_levenberg.cpp:62-68) and the plugin starts a new graph behind the same handle (wide seam:
    g2ohip_clear_edge_sets; narrow seam: the pattern is analysed again).  Both runs make progress, nothing fails."""
    pr = ba_case(10, 120)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr)
    out, _ = _run(host, prob, solver, 3, str(tmp_path / "t.json"), mode="twice")
    once, _ = _run(host, prob, solver, 3, str(tmp_path / "o.json"))
    assert out["iterations"] >= once["iterations"] + 1
    assert np.allclose(out["chi2"][:once["iterations"]], once["chi2"], rtol=1e-9, atol=0)
    assert out["chi2"][-1] <= out["chi2"][once["iterations"] - 1] * (1 + 1e-12)


@pytest.mark.parametrize("solver", ["gn_fix3_2_hip", "lm_fix3_2_hip", "lm_fix3_2_hipls"])
def test_planar_pose_graph_through_the_g2o_vtables(host, tmp_path, solver):
    """BASELINE.json config 1 (manhattan3500: VertexSE2 / EdgeSE2, BlockSolver_3_2 shape, no Schur complement) through the
    plugin: found by name, driven through the OptimizationAlgorithm -> Solver vtables.  The wide seam runs twice -- the
    device fast path (EdgeSE2 groups bound to g2ohip_pg_*: only the estimates cross PCIe) and the generic path (the host's
    linearizeOplus, Jacobians uploaded) -- and the narrow seam once; Gauss-Newton reproduces the reference's chi2 trajectory
    (golden vectors of the reference's own CSparse run), Levenberg-Marquardt descends monotonically towards the known optimum."""
    from tests.helpers import manhattan_golden
    g = manhattan_golden()
    path = str(tmp_path / "m.txt")
    with open(path, "w") as f:
        nv, ne = len(g["estimates"]), len(g["vi"])
        f.write("%d %d\n" % (nv, ne))
        for i in range(nv):
            f.write("%d %s\n" % (1 if g["hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in g["estimates"][i])))
        for k in range(ne):
            f.write("%d %d %s %s\n" % (g["vi"][k], g["vj"][k], " ".join("%.17g" % v for v in g["meas"][k]),
                                       " ".join("%.17g" % v for v in np.asarray(g["omega"][k]).reshape(-1))))
    runs = []
    for env in (({}, {"G2OHIP_ADAPTER_FASTPATH": "0"}) if solver.endswith("_hip") else ({},)):
        out, err = _run(host, path, solver, 8 if solver.startswith("lm") else 4, str(tmp_path / "o.json"), env, mode="se2")
        if solver.endswith("_hip"):
            assert ("device fast path for" in err) == (not env)
        runs.append(out)
    for out in runs:
        if solver.startswith("gn"):
            assert out["iterations"] == 4
            assert abs(out["chi2_initial"] - g["chi2_gn"][0]) <= 1e-6 * g["chi2_gn"][0]
            assert np.allclose(out["chi2"], g["chi2_gn"][1:5], rtol=1e-6, atol=0)
        else:
            assert out["chi2"][-1] < 0.05 * out["chi2_initial"] and all(b <= a * (1 + 1e-12) for a, b in zip(out["chi2"], out["chi2"][1:]))
            # (5 668 -> 175 in eight iterations; the optimum of this file is 146.08)
    if len(runs) == 2:      # fast path == generic path
        assert np.allclose(runs[0]["chi2"], runs[1]["chi2"], rtol=1e-9, atol=0)
        assert np.abs(np.array(runs[0]["poses"]) - np.array(runs[1]["poses"])).max() < 1e-7


def test_variable_shape_solver_names_read_the_shape_off_the_graph(host, tmp_path):
    """gn_var / lm_var / dl_var are the names g2o registers for BlockSolverX (solver_csparse.cpp:54-59) and what most g2o
    applications ask for.  `*_var_hip` / `*_var_hipdev` pick 3-2 / 6-3 / 7-3 from the dimensions of the graph's vertices when
    the algorithm initialises the solver and forward to the fixed-shape device solver: the SAME numbers as the fixed names, on a
    bundle-adjustment graph (6-3, host loop and device-resident driver) and on the planar pose graph (3-2, no marginalised
    vertex); the property reports variable dimensions (-1)."""
    pr = ba_case(30, 300)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr)
    for var, fix in (("lm_var_hip", "lm_fix6_3_hip"), ("lm_var_hipdev", "lm_fix6_3_hipdev"), ("dl_var_hip", "dl_fix6_3_hip")):
        a, erra = _run(host, prob, var, 4, str(tmp_path / "a.json"))
        b, _ = _run(host, prob, fix, 4, str(tmp_path / "b.json"))
        assert a["property"]["name"] == var and a["property"]["poseDim"] == -1 and a["property"]["landmarkDim"] == -1
        assert a["iterations"] == b["iterations"] and a["trials"] == b["trials"]
        assert a["chi2"] == b["chi2"] and a["lambda"] == b["lambda"] and a["cams"] == b["cams"] and a["points"] == b["points"]
        if var.endswith("hipdev"):
            assert DEV_ON in erra
    from tests.helpers import manhattan_golden
    g = manhattan_golden()
    path = str(tmp_path / "m.txt")
    with open(path, "w") as f:
        nv, ne = len(g["estimates"]), len(g["vi"])
        f.write("%d %d\n" % (nv, ne))
        for i in range(nv):
            f.write("%d %s\n" % (1 if g["hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in g["estimates"][i])))
        for k in range(ne):
            f.write("%d %d %s %s\n" % (g["vi"][k], g["vj"][k], " ".join("%.17g" % v for v in g["meas"][k]),
                                       " ".join("%.17g" % v for v in np.asarray(g["omega"][k]).reshape(-1))))
    a, _ = _run(host, path, "gn_var_hip", 4, str(tmp_path / "a.json"), mode="se2")
    b, _ = _run(host, path, "gn_fix3_2_hip", 4, str(tmp_path / "b.json"), mode="se2")
    assert a["chi2"] == b["chi2"] and a["poses"] == b["poses"]
    assert np.allclose(a["chi2"], g["chi2_gn"][1:5], rtol=1e-6, atol=0)


def test_3d_pose_graph_through_the_g2o_vtables(host, tmp_path):
    """BASELINE.json config 2 (sphere: VertexSE3 / EdgeSE3, BlockSolver_6_3 shape, no Schur complement) through the plugin
    and the vtables, Levenberg-Marquardt: on the device fast path (EdgeSE3 groups bound to g2ohip_pg_* type 2: analytic
    Jacobians of isometry3d_gradients.h restated on the device) and on the generic path (the test host's EdgeSE3 brings the
    same analytic Jacobians, uploaded).  chi2 of the initial guess and after the first damped step (lambda = 1e-5 max diag:
    computeLambdaInit) equal the golden trajectory of the reference's CSparse path; the two paths agree to rounding.  Then the
    WHOLE golden trajectory -- three damped steps at the fixture's fixed lambda -- through the Solver vtable on both paths."""
    from tests.helpers import sphere_golden
    g = sphere_golden()
    path = str(tmp_path / "s.txt")
    with open(path, "w") as f:
        nv, ne = len(g["poses"]), len(g["vi"])
        f.write("%d %d\n" % (nv, ne))
        for i in range(nv):
            f.write("%d %s\n" % (1 if g["hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in np.asarray(g["poses"][i]).reshape(-1))))
        for k in range(ne):
            f.write("%d %d %s %s\n" % (g["vi"][k], g["vj"][k], " ".join("%.17g" % v for v in np.asarray(g["Z"][k]).reshape(-1)),
                                       " ".join("%.17g" % v for v in np.asarray(g["omega"][k]).reshape(-1))))
    runs = []
    for env in ({}, {"G2OHIP_ADAPTER_FASTPATH": "0"}):
        out, err = _run(host, path, "lm_fix6_3_hip", 3, str(tmp_path / "o.json"), env, mode="se3")
        assert ("device fast path for" in err) == (not env)
        assert out["iterations"] == 3
        assert abs(out["chi2_initial"] - g["chi2_lm"][0]) <= 1e-6 * g["chi2_lm"][0]
        assert 0 < out["lambda"][0] < float(g["lambda0"])            # (accepted first step: lambda0 times at most 2/3, at least 1/3)
        assert all(b < a for a, b in zip([out["chi2_initial"]] + out["chi2"], out["chi2"]))
        runs.append(out)
    assert abs(runs[0]["chi2"][0] - g["chi2_lm"][1]) <= 1e-6 * g["chi2_lm"][1]        # analytic Jacobians on the device
    assert abs(runs[1]["chi2"][0] - g["chi2_lm"][1]) <= 1e-6 * g["chi2_lm"][1]        # ... and on the host
    assert np.allclose(runs[0]["chi2"], runs[1]["chi2"], rtol=1e-9, atol=0)
    assert relerr(runs[0]["poses"], runs[1]["poses"]) < 1e-9
    fixed = []
    for env in ({}, {"G2OHIP_ADAPTER_FASTPATH": "0"}):
        out, err = _run(host, path, "lm_fix6_3_hip", 2, str(tmp_path / "f.json"), env, mode="se3lambda:%.17g" % float(g["lambda0"]))
        assert ("device fast path for" in err) == (not env)
        assert out["iterations"] == 2
        traj = [out["chi2_initial"]] + out["chi2"]                                    # chi2 after 0, 1, 2 golden steps
        assert np.allclose(traj, g["chi2_lm"], rtol=1e-6, atol=0), (traj, g["chi2_lm"])
        fixed.append(out)
    assert np.allclose(fixed[0]["chi2"], fixed[1]["chi2"], rtol=1e-9, atol=0)


# ---- the device-resident drivers (openslam_g2o_amd/cpp/g2o_hip_algorithm.h): "<gn|lm>_fix<p>_<l>_hipdev"
DEV_ON = "device-resident iteration"
DEV_OFF = "g2o's host loop"


@pytest.mark.parametrize("huber", [0.0, 1.0])
def test_device_resident_levenberg_walks_the_host_loops_trajectory(host, tmp_path, huber):
    """lm_fix6_3_hipdev: OptimizationAlgorithmLevenbergHip keeps errors, chi2, oplus and the estimate stack on the device and
    writes the accepted estimates back into the vertices after every solve().  chi2 (evaluated by the HOST's
    computeActiveErrors on the written-back vertices after each iteration), lambda, the number of trials per iteration and
    the final estimates equal the oracle-driven loop and the host loop over the same Solver (lm_fix6_3_hip).  With the front
    end off (or G2OHIP_ADAPTER_DEVICE_LOOP=0) the same algorithm object runs g2o's host loop."""
    pr = ba_case(40, 400, outlier_frac=0.05 if huber > 0 else 0.0)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr, huber)
    chi0, n_o, chis_o, lams_o, trials_o, cams_o, pts_o = _oracle_lm(pr, 5, huber)
    ref, _ = _run(host, prob, "lm_fix6_3_hip", 5, str(tmp_path / "ref.json"))
    for tag, env, resident in (("device", {}, True), ("loop-off", {"G2OHIP_ADAPTER_DEVICE_LOOP": "0"}, False),
                               ("generic", {"G2OHIP_ADAPTER_FASTPATH": "0"}, False)):
        out, err = _run(host, prob, "lm_fix6_3_hipdev", 5, str(tmp_path / (tag + ".json")), env)
        assert out["property"] == {"name": "lm_fix6_3_hipdev", "type": "MI355X HIP", "requiresMarginalize": True, "poseDim": 6, "landmarkDim": 3}
        assert (DEV_ON in err) == resident and (DEV_OFF in err) == (not resident), err[-600:]
        assert abs(out["chi2_initial"] - chi0) <= 1e-9 * chi0
        assert out["iterations"] == n_o and out["trials"] == trials_o, (tag, out["trials"], trials_o)
        assert np.allclose(out["chi2"], chis_o, rtol=1e-7, atol=0), (tag, out["chi2"], chis_o)
        assert np.allclose(out["lambda"], lams_o, rtol=1e-7, atol=0), (tag, out["lambda"], lams_o)
        assert relerr(np.array(out["cams"]).reshape(-1, 12), cams_o) < 1e-7 and relerr(np.array(out["points"]).reshape(-1, 3), pts_o) < 1e-7
        # against the host loop over the same device solver: the same decisions on the same numbers
        assert out["trials"] == ref["trials"]
        assert np.allclose(out["chi2"], ref["chi2"], rtol=1e-9, atol=0) and np.allclose(out["lambda"], ref["lambda"], rtol=1e-9, atol=0)
        assert relerr(np.array(out["cams"]), np.array(ref["cams"])) < 1e-9 and relerr(np.array(out["points"]), np.array(ref["points"])) < 1e-9


@pytest.mark.parametrize("solver,huber", [("lm_fix6_3_hipdev", 0.0), ("lm_fix6_3_hipdev", 1.0), ("gn_fix6_3_hipdev", 0.0)])
def test_look_ahead_trial_of_the_device_resident_drivers_changes_no_number(host, tmp_path, solver, huber):
    """From iteration 1 on OptimizationAlgorithmLevenbergHip queues the head of the NEXT solve() (errors, buildSystem, push,
    setLambda, solve, update, errors, the trial's sums without a synchronisation) before it writes the accepted estimates into the
    vertices; the next solve() consumes that trial, anything else drops it (estimates popped).  The calls and their order are
    those of the run without look-ahead (G2OHIP_ADAPTER_LOOKAHEAD=0): chi2, lambda, trial counts and the final estimates are
    EQUAL, with accepted and with rejected look-ahead trials; the trial queued by the last iteration is dropped by the
    destructor, the one queued before a second optimize() by its iteration 0.  The Gauss-Newton driver queues its next iteration
    the same way (push, solve, update; the status comes back with the next solve())."""
    pr = ba_case(40, 400, outlier_frac=0.05 if huber > 0 else 0.0)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr, huber)
    for mode, n in ((None, 8), ("twice", 4)):
        off, err0 = _run(host, prob, solver, n, str(tmp_path / "off.json"), {"G2OHIP_ADAPTER_LOOKAHEAD": "0"}, mode=mode)
        on, err1 = _run(host, prob, solver, n, str(tmp_path / "on.json"), mode=mode)
        assert "look-ahead trials queued" not in err0
        line = [ln for ln in err1.splitlines() if "look-ahead trials queued" in ln]
        assert line, err1[-800:]
        queued, dropped = [int(w.strip(",")) for w in line[-1].split() if w.strip(",").isdigit()]
        assert queued >= 2 and 1 <= dropped <= (1 if mode is None else 2), line
        assert on["iterations"] == off["iterations"] and on["trials"] == off["trials"], (on["trials"], off["trials"])
        assert on["chi2"] == off["chi2"] and on["lambda"] == off["lambda"], (on["chi2"], off["chi2"])
        assert on["cams"] == off["cams"] and on["points"] == off["points"]


def test_device_resident_levenberg_uploads_again_on_a_second_optimize(host, tmp_path):
    """A second optimize() starts at iteration 0 again: the structure is rebuilt, the front ends are bound again and the
    vertices' estimates (the result of the first run) go up again -- the two runs chain like the host loop's."""
    pr = ba_case(30, 300)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr)
    ref, _ = _run(host, prob, "lm_fix6_3_hip", 3, str(tmp_path / "ref.json"), mode="twice")
    out, err = _run(host, prob, "lm_fix6_3_hipdev", 3, str(tmp_path / "dev.json"), mode="twice")
    assert err.count(DEV_ON) == 2
    assert out["iterations"] == ref["iterations"] == 6 and out["trials"] == ref["trials"]
    assert np.allclose(out["chi2"], ref["chi2"], rtol=1e-9, atol=0)
    assert relerr(np.array(out["cams"]), np.array(ref["cams"])) < 1e-9


def test_device_resident_drivers_fall_back_when_a_front_end_refuses(host, tmp_path):
    """A localisation graph (every point fixed) is refused by the BA front end: the edges go to generic groups and the
    device-resident driver runs g2o's host loop -- same trajectory as lm_fix6_3_hip."""
    pr = ba_case(20, 200)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr, fix_points=True)
    ref, _ = _run(host, prob, "lm_fix6_3_hip", 3, str(tmp_path / "ref.json"))
    out, err = _run(host, prob, "lm_fix6_3_hipdev", 3, str(tmp_path / "dev.json"))
    assert DEV_OFF in err and DEV_ON not in err
    assert out["trials"] == ref["trials"] and np.allclose(out["chi2"], ref["chi2"], rtol=1e-10, atol=0)


@pytest.mark.parametrize("solver", ["gn_fix3_2_hipdev", "lm_fix3_2_hipdev"])
def test_device_resident_drivers_on_the_planar_pose_graph(host, tmp_path, solver):
    """manhattan3500 (config 1) under the device-resident drivers: Gauss-Newton reproduces the golden chi2 trajectory of the
    reference's CSparse run, Levenberg-Marquardt the host loop's over the same solver; the poses written back into the
    vertices equal the host loop's."""
    from tests.helpers import manhattan_golden
    g = manhattan_golden()
    path = str(tmp_path / "m.txt")
    with open(path, "w") as f:
        nv, ne = len(g["estimates"]), len(g["vi"])
        f.write("%d %d\n" % (nv, ne))
        for i in range(nv):
            f.write("%d %s\n" % (1 if g["hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in g["estimates"][i])))
        for k in range(ne):
            f.write("%d %d %s %s\n" % (g["vi"][k], g["vj"][k], " ".join("%.17g" % v for v in g["meas"][k]),
                                       " ".join("%.17g" % v for v in np.asarray(g["omega"][k]).reshape(-1))))
    its = 8 if solver.startswith("lm") else 4
    ref, _ = _run(host, path, solver.replace("hipdev", "hip"), its, str(tmp_path / "r.json"), mode="se2")
    out, err = _run(host, path, solver, its, str(tmp_path / "o.json"), mode="se2")
    assert DEV_ON in err
    if solver.startswith("gn"):
        assert out["iterations"] == 4
        assert np.allclose(out["chi2"], g["chi2_gn"][1:5], rtol=1e-6, atol=0)
    assert out["iterations"] == ref["iterations"]
    assert np.allclose(out["chi2"], ref["chi2"], rtol=1e-9, atol=0)
    assert np.abs(np.array(out["poses"]) - np.array(ref["poses"])).max() < 1e-8


def test_device_resident_levenberg_on_the_3d_pose_graph(host, tmp_path):
    """The sphere (config 2) under lm_fix6_3_hipdev: chi2 of the initial guess and of the first damped step equal the golden
    trajectory of the reference's CSparse path; three iterations equal the host loop's over the same solver."""
    from tests.helpers import sphere_golden
    g = sphere_golden()
    path = str(tmp_path / "s.txt")
    with open(path, "w") as f:
        nv, ne = len(g["poses"]), len(g["vi"])
        f.write("%d %d\n" % (nv, ne))
        for i in range(nv):
            f.write("%d %s\n" % (1 if g["hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in np.asarray(g["poses"][i]).reshape(-1))))
        for k in range(ne):
            f.write("%d %d %s %s\n" % (g["vi"][k], g["vj"][k], " ".join("%.17g" % v for v in np.asarray(g["Z"][k]).reshape(-1)),
                                       " ".join("%.17g" % v for v in np.asarray(g["omega"][k]).reshape(-1))))
    ref, _ = _run(host, path, "lm_fix6_3_hip", 3, str(tmp_path / "r.json"), mode="se3")
    out, err = _run(host, path, "lm_fix6_3_hipdev", 3, str(tmp_path / "o.json"), mode="se3")
    assert DEV_ON in err and out["iterations"] == 3
    assert abs(out["chi2_initial"] - g["chi2_lm"][0]) <= 1e-6 * g["chi2_lm"][0]
    assert abs(out["chi2"][0] - g["chi2_lm"][1]) <= 1e-6 * g["chi2_lm"][1]
    assert np.allclose(out["chi2"], ref["chi2"], rtol=1e-9, atol=0) and np.allclose(out["lambda"], ref["lambda"], rtol=1e-9, atol=0)
    assert relerr(out["poses"], ref["poses"]) < 1e-9


def test_device_resident_levenberg_with_edge_classes_and_per_edge_kernels(host, tmp_path):
    """The graphs whose groups differ per EDGE -- two CameraParameters with Huber on every other edge (four BA edge classes),
    manhattan3500 with Huber on its loop closures only (one robust kernel per edge on the pose-graph front end) -- under the
    device-resident driver: chi2 on the device honours the classes / per-edge kernels, the trajectory is the host loop's."""
    from tests.helpers import manhattan_golden
    from tests.test_gpu_edge_classes import CLASSES
    pr = ba_case(40, 400, outlier_frac=0.05)
    second = (pr["cam_idx"] % 2) == 1
    robust = (np.arange(pr["E"]) % 2) == 1
    cls = (2 * second + robust).astype(np.int32)
    f0, c0 = pr["f"], np.array([pr["cx"], pr["cy"]])
    meas = (pr["meas"] - c0) / f0 * CLASSES[cls, 0][:, None] + CLASSES[cls, 1:3]
    prob = str(tmp_path / "p.txt")
    with open(prob, "w") as f:
        f.write("%d %d %d %.17g %.17g %.17g 0 %.17g %.17g %.17g\n" % (pr["P"], pr["L"], pr["E"], *CLASSES[0, :3], *CLASSES[2, :3]))
        for i in range(pr["P"]):
            f.write("%d %s\n" % (1 if pr["cam_hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in pr["cams"][i])))
        for j in range(pr["L"]):
            f.write("0 %s\n" % " ".join("%.17g" % v for v in pr["pts"][j]))
        for k in range(pr["E"]):
            f.write("%d %d %.17g %.17g %d %.17g\n" % (pr["cam_idx"][k], pr["pt_idx"][k], meas[k][0], meas[k][1], int(second[k]), CLASSES[cls[k], 4]))
    ref, _ = _run(host, prob, "lm_fix6_3_hip", 5, str(tmp_path / "r.json"), mode="classes")
    out, err = _run(host, prob, "lm_fix6_3_hipdev", 5, str(tmp_path / "o.json"), mode="classes")
    assert DEV_ON in err and "4 edge classes" in err
    assert out["trials"] == ref["trials"] and np.allclose(out["chi2"], ref["chi2"], rtol=1e-9, atol=0)
    assert np.allclose(out["lambda"], ref["lambda"], rtol=1e-9, atol=0) and relerr(np.array(out["cams"]), np.array(ref["cams"])) < 1e-9

    g = manhattan_golden()
    path = str(tmp_path / "m.txt")
    with open(path, "w") as f:
        nv, ne = len(g["estimates"]), len(g["vi"])
        f.write("%d %d\n" % (nv, ne))
        for i in range(nv):
            f.write("%d %s\n" % (1 if g["hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in g["estimates"][i])))
        for k in range(ne):
            f.write("%d %d %s %s\n" % (g["vi"][k], g["vj"][k], " ".join("%.17g" % v for v in g["meas"][k]),
                                       " ".join("%.17g" % v for v in np.asarray(g["omega"][k]).reshape(-1))))
    ref, _ = _run(host, path, "lm_fix3_2_hip", 6, str(tmp_path / "r2.json"), mode="se2huber:1.5")
    out, err = _run(host, path, "lm_fix3_2_hipdev", 6, str(tmp_path / "o2.json"), mode="se2huber:1.5")
    assert DEV_ON in err and out["iterations"] == 6
    assert np.allclose(out["chi2"], ref["chi2"], rtol=1e-9, atol=0)
    assert np.abs(np.array(out["poses"]) - np.array(ref["poses"])).max() < 1e-8


def test_device_resident_gauss_newton_through_online_growth(host, tmp_path):
    """updateInitialization + solve(iteration > 0) under gn_fix6_3_hipdev on the localisation graph of the online test: the
    graph is generic (fixed points), so the driver stays on g2o's host loop through the rebuilt structure and ends where
    gn_fix6_3_hip ends."""
    pr = ba_case(12, 150)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr, fix_points=True)
    ref, _ = _run(host, prob, "gn_fix6_3_hip", 2, str(tmp_path / "r.json"), mode="online:7")
    out, err = _run(host, prob, "gn_fix6_3_hipdev", 2, str(tmp_path / "o.json"), mode="online:7")
    assert DEV_OFF in err and DEV_ON not in err and out["iterations"] == ref["iterations"] == 4
    assert np.abs(np.array(out["cams"]) - np.array(ref["cams"])).max() < 1e-12


@pytest.mark.parametrize("mode", ["se2calib", "se2calib:1.5"])
def test_three_vertex_edges_through_the_g2o_vtables(host, tmp_path, mode):
    """BaseMultiEdge through the plugin: the odometry edges of manhattan3500 as THREE-vertex edges (pose, pose, one shared
    sensor-offset vertex: what g2o/types/sclam2d/edge_se2_sensor_calib.h computes; numeric Jacobians of BaseMultiEdge), the
    loop closures as EdgeSE2.  The wide seam registers one binary edge set per vertex pair of the three-vertex edges
    (g2ohip_set_edge_set_parts) and assembles on the device; the narrow seam assembles in the test host's own BlockSolver on
    the CPU and only factorises on the device: the two walk one Levenberg-Marquardt trajectory.  The second mode puts a
    Huber kernel on every third three-vertex edge (one robust kernel per edge on all three pair sets)."""
    from tests.helpers import manhattan_golden
    g = manhattan_golden()
    path = str(tmp_path / "m.txt")
    keep = 600                                               # the first 600 poses and the edges among them
    sel = [k for k in range(len(g["vi"])) if g["vi"][k] < keep and g["vj"][k] < keep]
    with open(path, "w") as f:
        f.write("%d %d\n" % (keep, len(sel)))
        for i in range(keep):
            f.write("%d %s\n" % (1 if g["hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in g["estimates"][i])))
        for k in sel:
            f.write("%d %d %s %s\n" % (g["vi"][k], g["vj"][k], " ".join("%.17g" % v for v in g["meas"][k]),
                                       " ".join("%.17g" % v for v in np.asarray(g["omega"][k]).reshape(-1))))
    wide, err = _run(host, path, "lm_fix3_2_hip", 6, str(tmp_path / "w.json"), mode=mode)
    narrow, _ = _run(host, path, "lm_fix3_2_hipls", 6, str(tmp_path / "n.json"), mode=mode)
    dev, errd = _run(host, path, "lm_fix3_2_hipdev", 6, str(tmp_path / "d.json"), mode=mode)
    assert DEV_OFF in errd                                   # (n-ary edges are generic: the device-resident driver runs the host loop)
    assert wide["iterations"] == narrow["iterations"] == 6
    assert abs(wide["chi2_initial"] - narrow["chi2_initial"]) <= 1e-12 * narrow["chi2_initial"]
    assert wide["chi2"][-1] < 0.2 * wide["chi2_initial"] and all(b <= a * (1 + 1e-12) for a, b in zip(wide["chi2"], wide["chi2"][1:]))
    # the first step: the same Jacobians (the host's, numeric) into two assemblies -- equal to rounding
    assert abs(wide["chi2"][0] - narrow["chi2"][0]) <= 1e-11 * narrow["chi2"][0], (wide["chi2"], narrow["chi2"])
    # later steps: the central differences (delta 1e-9: ~1e-7 of noise in a Jacobian) amplify the rounding differences of the two states
    assert np.allclose(wide["chi2"], narrow["chi2"], rtol=5e-6, atol=0), (wide["chi2"], narrow["chi2"])
    assert np.allclose(wide["lambda"], narrow["lambda"], rtol=1e-3, atol=0)
    assert np.abs(np.array(wide["poses"]) - np.array(narrow["poses"])).max() < 1e-3      # (poses of a 30 m trajectory, not converged yet)
    assert np.abs(np.array(wide["calib"]) - np.array(narrow["calib"])).max() < 1e-3
    assert np.allclose(dev["chi2"], wide["chi2"], rtol=1e-12, atol=0)
    assert np.abs(np.array(wide["calib"])).max() < 0.05      # (the file's odometry needs no offset: the calibration vertex goes back towards it)


@pytest.mark.parametrize("solver", ["gn_fix3_2_hipdev", "lm_fix3_2_hipdev"])
def test_device_resident_drivers_through_online_growth_of_a_pose_graph(host, tmp_path, solver):
    """Incremental SLAM under the device-resident drivers: manhattan3500 optimised for its first 3 000 poses, grown by the
    rest (SparseOptimizer::updateInitialization -> Solver::updateStructure: the structure is rebuilt, the pose-graph front end
    bound again, the device's estimates are gone), optimised on with iteration numbers > 0.  The driver has to notice that
    the device holds no estimates for the new structure and send the vertices' again; the run ends where the host loop over
    the same solver ends."""
    from tests.helpers import manhattan_golden
    g = manhattan_golden()
    path = str(tmp_path / "m.txt")
    with open(path, "w") as f:
        nv, ne = len(g["estimates"]), len(g["vi"])
        f.write("%d %d\n" % (nv, ne))
        for i in range(nv):
            f.write("%d %s\n" % (1 if g["hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in g["estimates"][i])))
        for k in range(ne):
            f.write("%d %d %s %s\n" % (g["vi"][k], g["vj"][k], " ".join("%.17g" % v for v in g["meas"][k]),
                                       " ".join("%.17g" % v for v in np.asarray(g["omega"][k]).reshape(-1))))
    ref, _ = _run(host, path, solver.replace("hipdev", "hip"), 3, str(tmp_path / "r.json"), mode="se2online:3000")
    out, err = _run(host, path, solver, 3, str(tmp_path / "o.json"), mode="se2online:3000")
    assert DEV_ON in err and DEV_OFF not in err
    assert out["iterations"] == ref["iterations"] == 6
    assert np.allclose(out["chi2"], ref["chi2"], rtol=1e-9, atol=0), (out["chi2"], ref["chi2"])
    assert np.abs(np.array(out["poses"]) - np.array(ref["poses"])).max() < 1e-8


def test_device_resident_levenberg_at_a_size_where_the_host_loops_are_threaded(host, tmp_path):
    """The estimate gather / write-back loops of the adapter run on several host threads from 8 192 vertices on
    (G2OHIP_ADAPTER_THREADS).  The host's bench mode (a synthetic band graph built in memory: 2 000 cameras, 20 000 points,
    100 000 observations) under lm_fix6_3_hipdev with 8 threads and with 1, and under the host loop: chi2 after every iteration --
    evaluated by the HOST on the written-back vertices -- is the same."""
    exe, plugin = host

    def run(solver, threads):
        out = str(tmp_path / ("%s_%d.json" % (solver, threads)))
        e = dict(os.environ, G2OHIP_ADAPTER_THREADS=str(threads))
        r = subprocess.run([exe, "none", plugin, solver, "4", out, "bench:2000:20000:5"], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        d = json.load(open(out))
        return d["chi2_initial"], [it["chi2"] for it in d["iterations"]]

    c0_ref, ref = run("lm_fix6_3_hip", 8)
    c0_a, a = run("lm_fix6_3_hipdev", 8)
    c0_b, b = run("lm_fix6_3_hipdev", 1)
    assert c0_ref == c0_a == c0_b and len(ref) == 4
    assert ref[-1] < 0.1 * c0_ref
    assert np.allclose(a, ref, rtol=1e-7, atol=0), (a, ref)     # (bench mode prints 9 digits)
    assert a == b


def test_threaded_edge_classification_equals_the_sequential_grouping(host, tmp_path):
    """From G2OHIP_ADAPTER_PAR_MIN edges on (default 200 000) the adapter classifies the EdgeProjectXYZ2UV edges on its host
    threads -- chunks of the edge list with their own class lists, merged in chunk order.  A graph of 15 000 observations with
    two CameraParameters and Huber on every other edge (four classes that every chunk meets in another order), forced through
    the threaded path, walks the trajectory of the sequential grouping, under the host loop and the device-resident driver."""
    from tests.test_gpu_edge_classes import CLASSES
    pr = ba_case(300, 3000, outlier_frac=0.05)
    second = ((pr["cam_idx"] // 7) % 2) == 1
    robust = (np.arange(pr["E"]) % 2) == 1
    cls = (2 * second + robust).astype(np.int32)
    f0, c0 = pr["f"], np.array([pr["cx"], pr["cy"]])
    meas = (pr["meas"] - c0) / f0 * CLASSES[cls, 0][:, None] + CLASSES[cls, 1:3]
    prob = str(tmp_path / "p.txt")
    with open(prob, "w") as f:
        f.write("%d %d %d %.17g %.17g %.17g 0 %.17g %.17g %.17g\n" % (pr["P"], pr["L"], pr["E"], *CLASSES[0, :3], *CLASSES[2, :3]))
        for i in range(pr["P"]):
            f.write("%d %s\n" % (1 if pr["cam_hidx"][i] < 0 else 0, " ".join("%.17g" % v for v in pr["cams"][i])))
        for j in range(pr["L"]):
            f.write("0 %s\n" % " ".join("%.17g" % v for v in pr["pts"][j]))
        for k in range(pr["E"]):
            f.write("%d %d %.17g %.17g %d %.17g\n" % (pr["cam_idx"][k], pr["pt_idx"][k], meas[k][0], meas[k][1], int(second[k]), CLASSES[cls[k], 4]))
    assert pr["E"] >= 8192
    for solver in ("lm_fix6_3_hip", "lm_fix6_3_hipdev"):
        seq, err0 = _run(host, prob, solver, 4, str(tmp_path / "s.json"), mode="classes")
        par, err1 = _run(host, prob, solver, 4, str(tmp_path / "t.json"), {"G2OHIP_ADAPTER_PAR_MIN": "1"}, mode="classes")
        assert "4 edge classes" in err0 and "4 edge classes" in err1
        assert par["trials"] == seq["trials"]
        assert np.allclose(par["chi2"], seq["chi2"], rtol=1e-10, atol=0) and np.allclose(par["lambda"], seq["lambda"], rtol=1e-10, atol=0)
        assert relerr(np.array(par["cams"]), np.array(seq["cams"])) < 1e-10


def test_hybrid_device_loop_with_a_prior_edge_no_front_end_knows(host, tmp_path):
    """One edge type outside the device front ends used to send lm_fix*_hipdev back to g2o's host loop for the WHOLE graph.  The
    hybrid loop keeps the bundle-adjustment group on the device and linearises only the foreign edges on the host (their vertices'
    trial estimates read back per trial, under push / pop): a user-defined unary prior on one camera (EdgeCameraPrior of the test
    host: BaseUnaryEdge, numeric Jacobian) next to 100 000 Huber-weighted observations walks the trajectory of the host loop
    (lm_fix6_3_hip: g2o's own Levenberg-Marquardt over the same solver), chi2 evaluated by the host on the written-back vertices;
    G2OHIP_ADAPTER_HYBRID=0 is the old behaviour (host loop, same trajectory)."""
    exe, plugin = host

    def run(solver, env=None):
        out = str(tmp_path / ("%s_%d.json" % (solver, len(env or {}))))      # (one file per run)
        e = dict(os.environ, G2OHIP_ADAPTER_VERBOSE="1")
        e.update(env or {})
        r = subprocess.run([exe, "none", plugin, solver, "5", out, "bench:2000:20000:5:prior:huber"], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        d = json.load(open(out))
        return d["chi2_initial"], [it["chi2"] for it in d["iterations"]], [it["levenbergIterations"] for it in d["iterations"]], r.stderr

    c0_ref, ref, tr_ref, _ = run("lm_fix6_3_hip")
    c0_h, hyb, tr_h, err_h = run("lm_fix6_3_hipdev")
    c0_o, off, tr_o, err_o = run("lm_fix6_3_hipdev", {"G2OHIP_ADAPTER_HYBRID": "0"})
    assert "hybrid device loop -- 1 host-linearised edges over 1 free vertices" in err_h and DEV_ON in err_h
    assert "hybrid device loop" not in err_o and DEV_OFF in err_o
    assert c0_ref == c0_h == c0_o and len(ref) == 5
    assert ref[-1] < 0.1 * c0_ref
    assert tr_h == tr_ref and tr_o == tr_ref
    assert np.allclose(hyb, ref, rtol=1e-7, atol=0), (hyb, ref)     # (bench mode prints 9 digits)
    assert np.allclose(off, ref, rtol=1e-7, atol=0)
    # the look-ahead trial under the hybrid loop: the host edges' trial errors are evaluated behind the write-back, the touched
    # vertices go back to the accepted estimates before solve() returns (the host's chi2 between the iterations sees them) and get
    # the trial's from the cache when it is accepted -- the numbers of the run without look-ahead
    assert "look-ahead trials queued" in err_h
    c0_n, nol, tr_n, err_n = run("lm_fix6_3_hipdev", {"G2OHIP_ADAPTER_LOOKAHEAD": "0", "G2OHIP_X": "1"})
    assert "look-ahead trials queued" not in err_n
    assert nol == hyb and tr_n == tr_h


def test_save_hessian_writes_the_octave_file_of_the_reference(host, tmp_path):
    """Solver::saveHessian (block_solver.hpp:628-632 -> SparseBlockMatrix::writeOctave(fileName, true), sparse_block_matrix.hpp:548-589)
    through the vtable: header, one-based "r c value" triplets of BOTH triangles sorted by column, and the values = Hpp of the last
    buildSystem -- rebuilt here by the oracle at the written-back estimates (LM ends an iteration on an accepted step, so the last
    system was linearised at the estimates of the iteration before: one iteration is run and compared at the INITIAL estimates)."""
    pr = ba_case(12, 80)
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr)
    hfile = str(tmp_path / "hpp.txt")
    out, err = _run(host, prob, "lm_fix6_3_hip", 1, str(tmp_path / "o.json"), {"G2OHIP_TEST_SAVE_HESSIAN": hfile})
    assert out["saveHessian"] is True
    lines = open(hfile).read().split("\n")
    assert lines[0] == "# name: " + hfile[:-4] and lines[1] == "# type: sparse matrix"
    nnz, rows, cols = (int(lines[k].split(": ")[1]) for k in (2, 3, 4))
    assert rows == cols == 6 * pr["nP"]
    trip = np.array([[float(v) for v in l.split()] for l in lines[6:] if l.strip()])
    assert len(trip) == nnz
    r, c = trip[:, 0].astype(int) - 1, trip[:, 1].astype(int) - 1
    assert np.all(np.diff(c * rows + r) > 0)                      # sorted by column, then row; no duplicates
    H = np.zeros((rows, cols))
    H[r, c] = trip[:, 2]
    assert np.array_equal(H, H.T)
    o = oracle_ba(pr)
    o.build_system()
    cp, ri = o.pattern("pp")
    V = o.values("Hpp").reshape(-1, 6, 6)
    Ho = np.zeros_like(H)
    for col in range(pr["nP"]):
        for q in range(cp[col], cp[col + 1]):
            Ho[ri[q] * 6:ri[q] * 6 + 6, col * 6:col * 6 + 6] = V[q].T
            Ho[col * 6:col * 6 + 6, ri[q] * 6:ri[q] * 6 + 6] = V[q]
    assert np.abs(H - Ho).max() <= 1e-8 * np.abs(Ho).max() + 5e-10      # (nine fixed digits in the file)


@pytest.mark.parametrize("solver", ["gn_fix6_3_hip", "gn_fix6_3_hipdev"])
def test_gauss_newton_on_a_numerically_singular_chain_repeats_the_solve_with_a_tiny_lambda(host, tmp_path, solver):
    """Gauss-Newton drop-in on long undamped camera chains (optimization_algorithm_gauss_newton.cpp:73-85): at 20 000 cameras the
    reduced system has kappa x eps >= 1; the reference's pivot test (csparse_helper.cpp:136) happens to pass under block-AMD and
    gn_fix6_3 takes the step, the nested-dissection factorisation meets d <= 0 (rounds 1-5: Fail at iteration 0).  The adapter
    now repeats an UNDAMPED solve that broke down once with lambda = 1e-14 x max diag, says so on cerr, and the iteration proceeds:
    both iterations return OK with finite chi2.  (What is NOT promised: a good step.  At kappa x eps >= 1 neither solver's increment
    has a correct digit -- chi2 goes 2.6e8 -> 3.5e9 -> 1.1e8 here --; the point is that optimize() runs on like gn_fix6_3's does.)"""
    exe, plugin = host
    out = str(tmp_path / "gn.json")
    r = subprocess.run([exe, "none", plugin, solver, "2", out, "bench:20000:100000:5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    assert "numerically singular" in r.stderr and "1e-14" in r.stderr
    d = json.load(open(out))
    chis = [it["chi2"] for it in d["iterations"]]
    assert len(chis) == 2 and np.isfinite(chis).all() and chis[-1] < d["chi2_initial"], (d["chi2_initial"], chis)


@pytest.mark.parametrize("solver", ["lm_fix6_3_hip", "lm_fix6_3_hipdev"])
def test_levenberg_through_the_plugin_on_the_grid_graph_with_visibility_by_distance(host, tmp_path, solver):
    """The graph that is not a camera trajectory (synthetic.make_ba_grid: 144 cameras on a lattice, ragged observation lists of 4-12,
    every camera coupled to ~40 others) through the g2o vtables, host loop and device-resident driver: chi2, lambda, the number of
    trials per iteration and the final estimates of five Levenberg-Marquardt iterations equal the oracle-driven loop."""
    from openslam_g2o_amd import synthetic as S
    pr = S.make_ba_grid(144)
    Jp, Jc, err = S.ba_linearize(pr)
    pr.update(Jp=Jp, Jc=Jc, err=err, omega=S.ba_omega(pr))
    prob = str(tmp_path / "p.txt")
    _write_problem(prob, pr, 1.0)
    chi0, n_o, chis_o, lams_o, trials_o, cams_o, pts_o = _oracle_lm(pr, 5, 1.0)
    out, err_ = _run(host, prob, solver, 5, str(tmp_path / "o.json"))
    assert abs(out["chi2_initial"] - chi0) <= 1e-9 * chi0
    assert out["iterations"] == n_o and out["trials"] == trials_o, (out["trials"], trials_o)
    assert np.allclose(out["chi2"], chis_o, rtol=1e-7, atol=0), (out["chi2"], chis_o)
    assert np.allclose(out["lambda"], lams_o, rtol=1e-7, atol=0)
    assert relerr(np.array(out["cams"]).reshape(-1, 12), cams_o) < 1e-7 and relerr(np.array(out["points"]).reshape(-1, 3), pts_o) < 1e-7
