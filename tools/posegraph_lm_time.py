#!/usr/bin/env python
"""Whole Levenberg-Marquardt iterations (linearise, build, damp, solve, update, chi2) on the two pose-graph fixtures with
the graph resident on the device: ms per LM iteration and the chi2 trajectory.  python tools/posegraph_lm_time.py [iterations]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openslam_g2o_amd import capi, lm
from tests.helpers import manhattan_golden, sphere_golden

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name in ("manhattan", "sphere"):
    g = manhattan_golden() if name == "manhattan" else sphere_golden()
    p, l, d, typ = (3, 2, 3, 1) if name == "manhattan" else (6, 3, 6, 2)
    est = g["estimates"] if name == "manhattan" else g["poses"]
    meas = g["meas"] if name == "manhattan" else g["Z"]
    def make():
        s = capi.HipBlockSolver(p, l, 0)
        k = s.addEdgeSet(d, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
        s.buildStructure(g["nP"], 0, False)
        s.pgSetEdges(k, typ, g["vi"], g["vj"], meas, g["omega"])
        s.pgSetEstimates(est, g["hidx"])
        s.setOption("use_graph", 1)
        return s, lm.DevicePoseGraph(s)
    s, gr = make()
    lm.optimize(gr, s, 3, "lm")                       # warm-up (graph capture, lazy analysis)
    s, gr = make()
    s.sync()
    t0 = time.perf_counter()
    done, chis, lams, trials = lm.optimize(gr, s, n_it, "lm")
    s.sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"graph": name, "lm_iterations": int(done), "lm_trials": [int(t) for t in trials], "ms_per_lm_iteration": 1e3 * dt / max(1, done),
                      "chi2_first": float(chis[0]), "chi2_last": float(chis[-1])}))
