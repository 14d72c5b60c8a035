#!/usr/bin/env python
"""One damped solve of a large loop-closure BA graph (synthetic.make_ba_loops): does the non-band path hold at scale?
python tools/probe/loops_scale.py [poses landmarks laps]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openslam_g2o_amd import lm, synthetic as S
P, L, laps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10000, 100000, 5)
t0 = time.perf_counter()
pr = S.make_ba_loops(P, L, laps=laps, hubs=int(os.environ.get("HUBS", "0")))
t1 = time.perf_counter()
s, g = lm.setup_device_ba(pr, huber_delta=1.0)
s.setOption("use_graph", float(os.environ.get("USE_GRAPH", "1")))
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("=")
    s.setOption(k, float(v))
t2 = time.perf_counter()
g.linearize(); chi0 = g.chi2()
s.buildSystem()
lam = 1e-4 * s.maxDiagonal()
s.setLambda(lam, True)
ok = s.solve()
t3 = time.perf_counter()
ts = []
for _ in range(5):
    a = time.perf_counter(); s.buildSystem(); s.setLambda(lam, True); ok = s.solve() and ok; s.restoreDiagonal(); ts.append(time.perf_counter() - a)
x, b = s.x(), s.b()
s.setLambda(lam, True)
r = s.multiplyHessian(x) - b
st = s.stats()
g.push(); g.update(); s.restoreDiagonal(); g.compute_active_errors()
print(json.dumps({"poses": pr["nP"], "landmarks": pr["nL"], "edges": pr["E"], "ok": bool(ok), "generate_s": t1 - t0, "setup_s": t2 - t1, "first_solve_s": t3 - t2,
                  "ms_per_iteration": 1e3 * min(ts), "residual_rel": float(np.abs(r).max() / np.abs(b).max()), "chi2": [chi0, g.chi2()],
                  "fronts": st["numFronts"], "levels": st["numLevels"], "maxFrontDim": st["maxFrontDim"], "choleskyNNZ": st["choleskyNNZ"]}))
