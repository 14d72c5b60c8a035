#!/usr/bin/env python
"""Soak of the in-launch hand-offs of the large-front path (per-front flags, ticketed partial sums): N repeated solves of the
sphere fixture and of a loop-closure BA graph must reproduce the first solution bit for bit.
python tools/probe/flag_soak.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openslam_g2o_amd import capi, synthetic as S
from oracle import oracle as O
from tests.helpers import sphere_golden

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = sphere_golden()
J0, J1, err = O.se3_edges(g["poses"], g["vi"], g["vj"], g["Z"])
for graph in (0, 1):
    s = capi.HipBlockSolver(6, 3, 0)
    k = s.addEdgeSet(6, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    s.buildStructure(g["nP"], 0, False)
    s.setEdgeData(k, J0, J1, g["omega"], err)
    s.setOption("use_graph", graph)
    s.buildSystem()
    lam = 1e-5 * s.maxDiagonal()
    bad = 0
    x0 = None
    for i in range(N):
        s.setLambda(lam, True); ok = s.solve(); s.restoreDiagonal()
        x = s.x()
        if x0 is None:
            x0 = x
        if not ok or not np.array_equal(x, x0):
            bad += 1
    print("sphere use_graph=%d: %d solves, %d differ, fallbacks %s" % (graph, N, bad, s.stats().get("dependencyFallbacks")))
pr = S.make_ba_loops(1500, 12000, laps=4, hubs=1)
Jp, Jc, err_ = S.ba_linearize(pr)
pr.update(Jp=Jp, Jc=Jc, err=err_, omega=S.ba_omega(pr))
from tests.helpers import hip_ba
s = hip_ba(pr)
s.buildSystem()
x0, bad = None, 0
for i in range(max(N // 5, 50)):
    s.setLambda(10.0, True); ok = s.solve(); s.restoreDiagonal()
    x = s.x()
    if x0 is None:
        x0 = x
    if not ok or not np.array_equal(x, x0):
        bad += 1
print("loop-closure BA: %d solves, %d differ, maxFrontDim %s fallbacks %s" % (max(N // 5, 50), bad, s.stats()["maxFrontDim"], s.stats().get("dependencyFallbacks")))
