#!/usr/bin/env python
"""The paths next to the headline number, one JSON line each (profiles/r5_paths.jsonl):
  * configs 1 and 2 of BASELINE.json (manhattan3500, sphere 2200, sphere2500; no Schur complement) -- one linear-solve
    iteration (errors + Jacobians, buildSystem, setLambda, solve, restoreDiagonal) on the MI355X through the device front
    end, and the CPU beside it ON THIS HOST, one thread: errors / Jacobians / buildSystem by the oracle (kind "port") and the
    linear solver by the REFERENCE's own compiled CSparse path (oracle/_ref: fillCCS -> cs_cholsolsymb, kind "reference";
    ordering + symbolic factorisation once, not timed, as LinearSolverCSparse caches them) -- BASELINE.md section 3;
  * bench.py --edge-data arrays, bench.py --information edge: run by tools/gpu_r5_paths.sh next to this script.
python tools/paths_bench.py [graphs]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openslam_g2o_amd import capi, synthetic as S
from oracle import oracle as O
from tests.helpers import manhattan_golden, sphere_golden


def host_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def med(v):
    return float(np.median(v))


CPU_ONLY = "--cpu-only" in sys.argv     # the CPU leg alone (build container: oracle/_ref = the reference's compiled CSparse is present there)
if CPU_ONLY:
    sys.argv.remove("--cpu-only")


def run_cpu_only(name, g, p, l, d, lin, v0, v1, reps_cpu=5):
    """errors / Jacobians / buildSystem by the oracle (port), the linear solver by the reference's own compiled CSparse path (oracle/_ref:
    block-AMD ordering + cs_schol once, then cs_cholsolsymb per iteration) -- no GPU involved."""
    import ctypes as C
    R = O.ref()
    assert R is not None, "oracle/_ref not built"
    o = O.OracleSolver(p, l, g["nP"], 0, schur=False)
    ko = o.add_edge_set(d, v0, v1)
    o.set_dims(ko, p, p)
    o.build_structure()
    J0, J1, err = lin()
    o.set_edge_data(ko, J0, J1, g["omega"], err)
    o.build_system()
    lam = 1e-5 * o.max_diagonal()
    ip = lambda a: a.ctypes.data_as(O.c_int_p)
    dp = lambda a: a.ctypes.data_as(O.c_dbl_p)
    t_lin, t_asm, t_ref, t_port = [], [], [], []
    h = None
    for rep in range(reps_cpu + 1):
        t0 = time.perf_counter(); J0, J1, err = lin(); tl = time.perf_counter() - t0
        o.set_edge_data(ko, J0, J1, g["omega"], err)
        t0 = time.perf_counter(); o.build_system(); ta = time.perf_counter() - t0
        o.set_lambda(lam, True)
        t0 = time.perf_counter(); assert o.solve(); tp = time.perf_counter() - t0
        xo = o.x()
        cp, ri = o.pattern("pp")
        cp, ri = np.ascontiguousarray(cp, np.int32), np.ascontiguousarray(ri, np.int32)
        nb, n = g["nP"], g["nP"] * p
        if h is None:
            P = np.zeros(nb, np.int32)
            t0 = time.perf_counter()
            Ap, Ai, Ax = O.scalar_ccs(nb, p, cp, ri, o.values("Hpp"))
            assert R.ref_block_amd(nb, ip(cp), ip(ri), ip(P))
            sperm = (P[:, None] * p + np.arange(p, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)
            h = C.c_void_p(R.ref_symbolic(n, ip(Ap), ip(Ai), ip(sperm)))
            t_sym = time.perf_counter() - t0
        t0 = time.perf_counter()
        Ap, Ai, Ax = O.scalar_ccs(nb, p, cp, ri, o.values("Hpp"))
        xr = o.b().copy()
        ok = R.ref_cholsolve(h, ip(Ap), ip(Ai), dp(Ax), dp(xr))
        tr = time.perf_counter() - t0
        assert ok
        o.restore_diagonal()
        if rep > 0:
            t_lin.append(tl); t_asm.append(ta); t_port.append(tp); t_ref.append(tr)
    print(json.dumps({"path": "CPU leg of config %s (%s), build container" % ("1" if name == "manhattan" else "2", name), "graph": name,
                      "poses": int(g["nP"]), "edges": int(len(g["vi"])), "kind": "port (assembly) + reference (linear solver)", "cores": 1,
                      "host": host_model(), "host_cores": os.cpu_count(), "linearize_ms": 1e3 * med(t_lin), "build_system_ms": 1e3 * med(t_asm),
                      "reference_solve_ms": 1e3 * med(t_ref), "reference_ordering_symbolic_ms": 1e3 * t_sym, "reference_lnz": R.ref_lnz(h),
                      "port_solve_ms": 1e3 * med(t_port), "value_ms_per_iteration": 1e3 * (med(t_lin) + med(t_asm) + med(t_ref)),
                      "dx_rel_err_reference_vs_port": float(np.abs(xr - xo).max() / np.abs(xo).max())}), flush=True)
    R.ref_free(h)


def run(name, reps_gpu=50, reps_cpu=5):
    if name == "manhattan":
        g = manhattan_golden(); p, l, d, typ = 3, 2, 3, 1
        est, meas = g["estimates"], g["meas"]
        lin = lambda: O.se2_edges(est, g["vi"], g["vj"], meas)
    else:
        g = S.make_sphere() if name == "sphere2500" else sphere_golden(); p, l, d, typ = 6, 3, 6, 2
        est, meas = g["poses"], g["Z"]
        lin = lambda: O.se3_edges(est, g["vi"], g["vj"], meas)
    v0, v1 = g["hidx"][g["vi"]], g["hidx"][g["vj"]]
    if CPU_ONLY:
        return run_cpu_only(name, g, p, l, d, lin, v0, v1)
    # ---- MI355X: the device front end (errors + Jacobians evaluated inside the iteration, like the reference's buildSystem)
    s = capi.HipBlockSolver(p, l, 0)
    k = s.addEdgeSet(d, v0, v1)
    s.buildStructure(g["nP"], 0, False)
    s.pgSetEdges(k, typ, g["vi"], g["vj"], meas, g["omega"])
    s.pgSetEstimates(est, g["hidx"])
    s.setOption("use_graph", 1)
    s.pgLinearize(True)
    s.buildSystem()
    lam = 1e-5 * s.maxDiagonal()

    def step():
        s.pgLinearize(True)
        s.buildSystem()
        s.setLambda(lam, True)
        ok = s.solve()
        s.restoreDiagonal()
        return ok
    for _ in range(5):
        assert step()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(reps_gpu):
        step()
    s.sync()
    gpu_iter = (time.perf_counter() - t0) / reps_gpu
    t0 = time.perf_counter()
    for _ in range(reps_gpu):
        s.setLambda(lam, True); s.solve(); s.restoreDiagonal()
    s.sync()
    gpu_solve = (time.perf_counter() - t0) / reps_gpu
    s.setLambda(lam, True); assert s.solve(); xg = s.x(); s.restoreDiagonal()
    st = s.stats()
    # ---- CPU on this host, one thread
    o = O.OracleSolver(p, l, g["nP"], 0, schur=False)
    ko = o.add_edge_set(d, v0, v1)
    o.set_dims(ko, p, p)
    o.build_structure()
    t_lin, t_asm, t_ref, t_port = [], [], [], []
    R = O.ref()
    for rep in range(reps_cpu + 1):
        t0 = time.perf_counter(); J0, J1, err = lin(); tl = time.perf_counter() - t0
        o.set_edge_data(ko, J0, J1, g["omega"], err)
        t0 = time.perf_counter(); o.build_system(); ta = time.perf_counter() - t0
        o.set_lambda(lam, True)
        t0 = time.perf_counter(); assert o.solve(); tp = time.perf_counter() - t0     # the port's own solve (rep 0 carries its ordering)
        xo = o.x()
        if rep == 0 and R is not None:
            # the reference's own compiled CSparse path on the damped Hpp: ordering + symbolic once (not timed)
            cp, ri = o.pattern("pp")
            cp, ri = np.ascontiguousarray(cp, np.int32), np.ascontiguousarray(ri, np.int32)
            nb, n = g["nP"], g["nP"] * p
            P = np.zeros(nb, np.int32)
            ip = lambda a: a.ctypes.data_as(O.c_int_p)
            dp = lambda a: a.ctypes.data_as(O.c_dbl_p)
            t0 = time.perf_counter()
            Ap, Ai, Ax = O.scalar_ccs(nb, p, cp, ri, o.values("Hpp"))
            assert R.ref_block_amd(nb, ip(cp), ip(ri), ip(P))
            sperm = (P[:, None] * p + np.arange(p, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)
            import ctypes as C
            h = C.c_void_p(R.ref_symbolic(n, ip(Ap), ip(Ai), ip(sperm)))
            t_sym = time.perf_counter() - t0
            lnz_ref = R.ref_lnz(h)
        if R is not None:
            t0 = time.perf_counter()
            Ap, Ai, Ax = O.scalar_ccs(nb, p, cp, ri, o.values("Hpp"))     # fillCCS of the damped matrix: part of LinearSolverCSparse::solve
            xr = o.b().copy()
            ok = R.ref_cholsolve(h, ip(Ap), ip(Ai), dp(Ax), dp(xr))
            tr = time.perf_counter() - t0
            assert ok
        o.restore_diagonal()
        if rep > 0:
            t_lin.append(tl); t_asm.append(ta); t_port.append(tp)
            if R is not None:
                t_ref.append(tr)
    row = {"path": "config %s: %s pose graph, no Schur complement" % ("1" if name == "manhattan" else "2", name), "graph": name,
           "poses": int(g["nP"]), "edges": int(len(g["vi"])), "block": p,
           "gpu_ms_per_iteration": 1e3 * gpu_iter, "gpu_ms_per_solve_only": 1e3 * gpu_solve,
           "gpu_fronts": st["numFronts"], "gpu_levels": st["numLevels"], "gpu_maxFrontDim": st["maxFrontDim"], "gpu_choleskyNNZ": st["choleskyNNZ"],
           "dx_rel_err_vs_port": float(np.abs(xg - xo).max() / np.abs(xo).max()),
           "cpu_baseline": {"unit": "ms/iter", "cores": 1, "host": host_model(), "host_cores": os.cpu_count(),
                            "sample": "median of %d iterations; errors + Jacobians and buildSystem by the oracle (port), linear solver %s" % (
                                reps_cpu, "by the reference's own compiled CSparse code (oracle/_ref: fillCCS + cs_cholsolsymb; cs_amd block ordering + symbolic once, not timed)"
                                if R is not None else "by the oracle's restatement (oracle/_ref absent)"),
                            "linearize_ms": 1e3 * med(t_lin), "build_system_ms": 1e3 * med(t_asm),
                            "port_solve_ms": 1e3 * med(t_port)}}
    cb = row["cpu_baseline"]
    if R is not None:
        cb.update(kind="port (assembly) + reference (linear solver)", reference_solve_ms=1e3 * med(t_ref), reference_ordering_symbolic_ms=1e3 * t_sym,
                  reference_lnz=lnz_ref, dx_rel_err_vs_reference=float(np.abs(xg - xr).max() / np.abs(xr).max()),
                  value=1e3 * (med(t_lin) + med(t_asm) + med(t_ref)))
        R.ref_free(h)
    else:
        cb.update(kind="port", value=1e3 * (med(t_lin) + med(t_asm) + med(t_port)))
    row["speedup_vs_cpu"] = cb["value"] / row["gpu_ms_per_iteration"]
    print(json.dumps(row), flush=True)


for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("manhattan", "sphere", "sphere2500")):
    run(name)
