"""Repeatability soak of the scratch-slab path on the grid graph (big_panel_solve_kernel, the writing extend-add, grouped updates):
N factorisations + solves of the same 10 000-camera system must give the same bits; prints the number of distinct results."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openslam_g2o_amd import synthetic as S, lm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pr = S.make_ba_grid(int(sys.argv[2]) if len(sys.argv) > 2 else 10000)
s, g = lm.setup_device_ba(pr, huber_delta=1.0)
g.linearize()
seen = {}
for i in range(n):
    s.buildSystem()
    s.setLambda(1e-5 * s.maxDiagonal(), True)
    assert s.solve()
    h = hashlib.sha1(np.ascontiguousarray(s.x()).tobytes()).hexdigest()
    seen[h] = seen.get(h, 0) + 1
    s.restoreDiagonal()
x = s.x()
print("solves", n, "distinct results", len(seen), seen)
