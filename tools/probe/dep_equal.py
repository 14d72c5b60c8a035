import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.helpers import ba_case, hip_ba
pr = ba_case(600, 6000)
xs = []
names = []
for opts in ({"dep_levels": 0}, {"dep_levels": 0}, {"dep_levels": 16}, {"dep_levels": 16, "dep_backward": 0},
             {"dep_levels": 16, "dep_spin_limit": 0}, {"dep_levels": 3, "use_graph": 1}, {"dep_levels": 0, "fuse_schur_reduce": 0}):
    s = hip_ba(pr, options=opts)
    for it in range(3):
        s.buildSystem(); s.setLambda(10.0, True); assert s.solve(); s.restoreDiagonal()
        xs.append(s.x()); names.append((str(opts), it))
for n, x in zip(names, xs):
    print(n, np.array_equal(x, xs[0]), float(np.abs(x - xs[0]).max()))
