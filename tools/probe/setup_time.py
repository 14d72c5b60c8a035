#!/usr/bin/env python
"""Where the one-time set-up of the metric configuration goes (host side): buildStructure (contributor lists, Schur pattern, tiles,
nested dissection + symbolic factorisation), baSetEdges (per-observation copies, slot tables), first solve (code loading, graph capture)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openslam_g2o_amd import capi, synthetic as S
P, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 1000000)
pr = S.make_ba_problem(P, L)
t = [time.perf_counter()]
s = capi.HipBlockSolver(6, 3, 0)
k = s.addEdgeSet(2, pr["v0"], pr["v1"]); t.append(time.perf_counter())
s.buildStructure(pr["nP"], pr["nL"], True); t.append(time.perf_counter())
s.baSetEdges(k, pr["cam_idx"], pr["pt_idx"], pr["meas"], None, pr["f"], pr["cx"], pr["cy"]); t.append(time.perf_counter())
s.baSetEstimates(pr["cams"], pr["cam_hidx"], pr["pts"], np.arange(pr["L"], dtype=np.int32)); t.append(time.perf_counter())
s.baLinearize(True); s.buildSystem(); s.setLambda(10.0, True); ok = s.solve(); s.restoreDiagonal(); t.append(time.perf_counter())
s.buildSystem(); s.setLambda(10.0, True); ok = s.solve(); s.restoreDiagonal(); t.append(time.perf_counter())
st = s.stats()
names = ["create+addEdgeSet", "buildStructure", "baSetEdges", "baSetEstimates", "first build+solve", "second build+solve"]
print(json.dumps({"poses": P, "landmarks": L, **{n: round(t[i + 1] - t[i], 4) for i, n in enumerate(names)}, "timeSymbolicDecomposition": st["timeSymbolicDecomposition"]}))
