"""Race soak of the round-6 wave_front_kernel (LDS regions reused between phases, five barriers per front, late L-panel stores):
N damped solves of the metric configuration must be bit-identical, with the damping changed in between (other numbers through the
same LDS), and the L panels must reproduce the solution through a separate backward pass.  python tools/probe/soak_wave_front.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openslam_g2o_amd import lm, synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pr = S.make_ba_problem(100000, 1000000)
s, g = lm.setup_device_ba(pr, options={"use_graph": 1})
g.linearize(); s.buildSystem()
lam = 1e-5 * s.maxDiagonal()
ref = {}
bad = 0
for it in range(N):
    l = lam * (1.0 + (it % 3))
    s.setLambda(l, True)
    assert s.solve()
    x = s.x()
    s.restoreDiagonal()
    k = it % 3
    if k not in ref:
        ref[k] = x
        r = s.multiplyHessian(x) - s.b()      # (undamped product: only a sanity figure)
    elif not np.array_equal(x, ref[k]):
        bad += 1
        print("iteration", it, "differs: max", np.abs(x - ref[k]).max())
print("soak: %d solves, %d mismatches" % (N, bad))
