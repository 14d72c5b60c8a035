import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.helpers import ba_case, hip_ba, relerr
pr = ba_case(120, 1500)
xs = {}
for ls in (0, 1, 2):
    s = hip_ba(pr, options={"linear_solver": ls, "pcg_tolerance": 1e-20, "pcg_absolute_tolerance": 0, "pcg_max_iterations": 6000})
    s.buildSystem(); s.setLambda(5.0, True)
    ok = s.solve(); x = s.x(); xs[ls] = x
    r = s.multiplyHessian(x) - s.b()
    print(ls, ok, s.stats()["iterationsLinearSolver"], np.abs(r).max() / np.abs(s.b()).max(), relerr(x, xs[0]), relerr(x[:714], xs[0][:714]))
