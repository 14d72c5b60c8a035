#!/usr/bin/env python
"""Where the FIRST Levenberg-Marquardt iteration of config 5 goes (lm_bench.py: 5.5 ms against 1.35 ms in steady state): every call of the
loop followed by a synchronisation, for the first three iterations."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openslam_g2o_amd import lm, synthetic as S
pr = S.make_ba_problem(100000, 1000000, outlier_frac=0.05)
s, g = lm.setup_device_ba(pr, huber_delta=1.0)
g.compute_active_errors(); g.chi2(); s.sync()
rows = []
lam = None
for it in range(3):
    t = {}
    def lap(name, fn):
        a = time.perf_counter(); r = fn(); s.sync(); t[name] = round(1e3 * (time.perf_counter() - a), 3); return r
    lap("linearize", g.linearize); lap("chi2", g.chi2); lap("buildSystem", s.buildSystem)
    if it == 0:
        lam = 1e-5 * lap("maxDiagonal", s.maxDiagonal)
    lap("push", g.push); lap("setLambda", lambda: s.setLambda(lam, True)); lap("solveAsync", s.solveAsync); lap("update", g.update)
    lap("restoreDiagonal", s.restoreDiagonal); lap("compute_active_errors", g.compute_active_errors); lap("trialStats", lambda: s.trialStats(lam))
    lap("discard_top", g.discard_top)
    t["sum"] = round(sum(t.values()), 3)
    rows.append(t)
print(json.dumps(rows))
