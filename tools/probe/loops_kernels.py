#!/usr/bin/env python
"""Per-kernel times of one iteration of a loop-closure BA graph (synthetic.make_ba_loops): where the non-band path spends its time.
python tools/probe/loops_kernels.py [poses landmarks laps]   (HUBS, OPTS="name=value ..." from the environment)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openslam_g2o_amd import lm, synthetic as S
P, L, laps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10000, 100000, 5)
pr = S.make_ba_loops(P, L, laps=laps, hubs=int(os.environ.get("HUBS", "0")))
s, g = lm.setup_device_ba(pr, huber_delta=1.0)
s.setOption("use_graph", float(os.environ.get("USE_GRAPH", "1")))
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("=")
    s.setOption(k, float(v))
g.linearize()
s.buildSystem()
lam = 1e-4 * s.maxDiagonal()
for _ in range(3):
    s.buildSystem(); s.setLambda(lam, True); assert s.solve(); s.restoreDiagonal()
s.setProfiling(True)
s.kernelTimes(reset=True)
n = 5
for _ in range(n):
    s.buildSystem(); s.setLambda(lam, True); assert s.solve(); s.restoreDiagonal()
kt = s.kernelTimes(reset=True)
st = s.stats()
out = {k: {"ms_per_iteration": 1e3 * v[0] / n, "launches_per_iteration": v[1] / n} for k, v in kt.items() if v[1]}
print(json.dumps({"poses": pr["nP"], "landmarks": pr["nL"], "edges": pr["E"], "levels": st["numLevels"], "fronts": st["numFronts"],
                  "maxFrontDim": st["maxFrontDim"], "choleskyNNZ": st["choleskyNNZ"], "kernels": out}))
