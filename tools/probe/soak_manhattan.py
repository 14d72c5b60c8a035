#!/usr/bin/env python
"""Soak of the dependency-driven launch that holds ALL LDS levels of a pose graph (dep_levels = 64: 26 levels of the manhattan
fixture in one launch): N solves must reproduce the first solution bit for bit, no stall fallback.
python tools/probe/soak_manhattan.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openslam_g2o_amd import capi
from oracle import oracle as O
from tests.helpers import manhattan_golden

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = manhattan_golden()
J0, J1, err = O.se2_edges(g["estimates"], g["vi"], g["vj"], g["meas"])
for graph in (0, 1):
    s = capi.HipBlockSolver(3, 2, 0)
    k = s.addEdgeSet(3, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    s.buildStructure(g["nP"], 0, False)
    s.setEdgeData(k, J0, J1, g["omega"], err)
    s.setOption("use_graph", graph)
    s.buildSystem()
    lam = float(g["lambda0"])
    x0, bad = None, 0
    for i in range(N):
        s.setLambda(lam, True); ok = s.solve(); s.restoreDiagonal()
        x = s.x()
        if x0 is None:
            x0 = x
        if not ok or not np.array_equal(x, x0):
            bad += 1
    print("manhattan use_graph=%d: %d solves, %d differ, fallbacks %d" % (graph, N, bad, s.stats()["dependencyFallbacks"]))
