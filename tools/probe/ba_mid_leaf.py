import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from openslam_g2o_amd import lm, synthetic as S
for P, L in ((2000, 20000), (20000, 200000)):
    for leaf in (32, 8, 4):
        pr = S.make_ba_problem(P, L)
        s, g = lm.setup_device_ba(pr, options={"nd_leaf": leaf, "use_graph": 1})
        g.linearize(); s.buildSystem()
        for _ in range(3):
            s.setLambda(10.0, True); assert s.solve(); s.restoreDiagonal()
        s.sync(); t0 = time.perf_counter()
        for _ in range(30):
            s.setLambda(10.0, True); s.solve(); s.restoreDiagonal()
        s.sync(); dt = (time.perf_counter() - t0) / 30
        st = s.stats()
        print("BA P=%d nd_leaf=%d: %.3f ms/solve fronts %d levels %d nnz %d band %d" % (P, leaf, 1e3 * dt, st["numFronts"], st["numLevels"], st["choleskyNNZ"], st["bandChains"]))
