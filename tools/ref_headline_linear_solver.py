#!/usr/bin/env python
"""The reference's OWN compiled linear solver (oracle/_ref: the reference's EXTERNAL/csparse + csparse_helper.cpp, i.e. what
LinearSolverCSparse::solve runs, linear_solver_csparse.h:106-142) timed on the reduced system of the HEADLINE workload
(100 000 poses / 1 000 000 landmarks, BASELINE.json configs[3]) in the BUILD container -- the GPU box cannot rebuild
oracle/_ref (no /root/reference there), so this number is recorded in BASELINE.md with its host instead of travelling.
The oracle (port) assembles the system and forms Hschur; the reference code then does fillCCS -> cs_cholsolsymb with the
block-AMD ordering + symbolic factorisation cached (not timed), exactly like LinearSolverCSparse between init() calls.
  python tools/ref_headline_linear_solver.py [poses landmarks]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openslam_g2o_amd import synthetic as S
from oracle import oracle as O

P, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 1000000)
R = O.ref()
assert R is not None, "oracle/_ref not built (make -C oracle in a container that has /root/reference)"
prob = S.make_ba_problem(P, L)
prob["omega"] = S.ba_omega(prob)
Jp, Jc, err = O.ba_edges(prob["cams"], prob["pts"], prob["cam_idx"], prob["pt_idx"], prob["meas"], prob["f"], prob["cx"], prob["cy"])
o = O.OracleSolver(6, 3, prob["nP"], prob["nL"], True)
k = o.add_edge_set(2, prob["v0"], prob["v1"])
o.set_dims(k, 3, 6)
o.build_structure()
o.set_edge_data(k, Jp, Jc, prob["omega"], err)
o.build_system()
lam = 1e-5 * 1.0e6
o.set_lambda(lam, True)
t0 = time.perf_counter(); o.solve_schur(); t_schur = time.perf_counter() - t0
t0 = time.perf_counter(); assert o.solve_reduced(); t_port = time.perf_counter() - t0     # the port's own reduced solve (carries its ordering the first time)
t0 = time.perf_counter(); assert o.solve_reduced(); t_port = time.perf_counter() - t0
xp_port = o.x()[:6 * prob["nP"]].copy()
cp, ri = o.pattern("hs")
cp, ri = np.ascontiguousarray(cp, np.int32), np.ascontiguousarray(ri, np.int32)
nb, n, p = prob["nP"], prob["nP"] * 6, 6
ip = lambda a: a.ctypes.data_as(O.c_int_p)
dp = lambda a: a.ctypes.data_as(O.c_dbl_p)
perm = np.zeros(nb, np.int32)
t0 = time.perf_counter()
Ap, Ai, Ax = O.scalar_ccs(nb, p, cp, ri, o.values("Hschur"))
assert R.ref_block_amd(nb, ip(cp), ip(ri), ip(perm))
sperm = (perm[:, None] * p + np.arange(p, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)
h = C.c_void_p(R.ref_symbolic(n, ip(Ap), ip(Ai), ip(sperm)))
t_sym = time.perf_counter() - t0
ts, tf = [], []
for rep in range(4):
    t0 = time.perf_counter()
    Ap, Ai, Ax = O.scalar_ccs(nb, p, cp, ri, o.values("Hschur"))      # fillCCS (sparse_block_matrix_ccs.h:143-199) restated by the oracle: timed apart
    tf.append(time.perf_counter() - t0)
    xr = o.bschur().copy()
    t0 = time.perf_counter()
    ok = R.ref_cholsolve(h, ip(Ap), ip(Ai), dp(Ax), dp(xr))           # the reference's compiled cs_cholsolsymb: numeric Cholesky + two triangular solves
    ts.append(time.perf_counter() - t0)
    assert ok
host = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
print(json.dumps({"workload": "reduced system of the synthetic BA %d poses / %d landmarks, lambda = %g" % (P, L, lam),
                  "reference_linear_solver_ms": 1e3 * float(np.median(ts[1:])), "fill_ccs_ms(oracle restatement, not the reference code)": 1e3 * float(np.median(tf[1:])), "kind": "reference", "cores": 1,
                  "what": "oracle/_ref: csparse_extension::cs_cholsolsymb on the scalar CCS matrix (block-AMD ordering + cs_schol symbolic once, not timed: %.0f ms)" % (1e3 * t_sym),
                  "reference_lnz": int(R.ref_lnz(h)), "port_reduced_solve_ms": 1e3 * t_port, "port_schur_ms": 1e3 * t_schur,
                  "dx_pose_rel_err_reference_vs_port": float(np.abs(xr - xp_port).max() / np.abs(xp_port).max()),
                  "host": host[0] if host else "unknown", "host_cores": os.cpu_count(), "where": "build container (not the GPU box)"}))
R.ref_free(h)
