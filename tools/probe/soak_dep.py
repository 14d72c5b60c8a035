"""Soak test of the dependency-driven launches: many BA problems of random size, every solve compared bit for bit
(pose part of x after a per-level backward sweep) with the one-launch-per-level factorisation; repeated solves per problem."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.helpers import ba_case, hip_ba

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_prob = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for it in range(n_prob):
    P = int(rng.integers(40, int(os.environ.get("SOAK_PMAX", "1500"))))
    L = int(P * rng.integers(4, 12))
    K = int(rng.integers(3, 9))
    leaf = int(rng.choice([4, 8, 16, 32]))
    pr = ba_case(P, L, seed=int(rng.integers(1, 1 << 30)), obs_per_landmark=K)
    ref = hip_ba(pr, options={"dep_levels": 0, "nd_leaf": leaf})
    dep = hip_ba(pr, options={"dep_levels": 16, "dep_backward": 0, "nd_leaf": leaf, "use_graph": int(rng.integers(0, 2))})
    depb = hip_ba(pr, options={"dep_levels": 16, "nd_leaf": leaf})
    xs = []
    for s in (ref, dep, depb):
        s.buildSystem()
        s.setLambda(1.0, True)
        ok = s.solve()
        xs.append((ok, s.x()))
        for rep in range(3):
            s.restoreDiagonal(); s.buildSystem(); s.setLambda(1.0, True)
            ok2 = s.solve()
            if ok2 != ok or not np.array_equal(s.x(), xs[-1][1]):
                bad += 1
                print("NOT REPEATABLE", it, P, L, K, leaf)
    if not (xs[0][0] and xs[1][0] and xs[2][0]):
        print("solve failed", it, P, L, K, leaf, [x[0] for x in xs]); bad += 1; continue
    if not np.array_equal(xs[0][1], xs[1][1]):
        bad += 1
        print("MISMATCH factor", it, P, L, K, leaf, float(np.abs(xs[0][1] - xs[1][1]).max()))
    e = float(np.abs(xs[0][1] - xs[2][1]).max() / np.abs(xs[0][1]).max())
    if e > 1e-12:
        bad += 1
        print("MISMATCH backward", it, P, L, K, leaf, e)
print("problems", n_prob, "bad", bad)
