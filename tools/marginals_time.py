#!/usr/bin/env python
"""Time of computeMarginals for ALL diagonal pose blocks of the metric configuration (100 000 poses): sparse-inverse
recursion against a sample of the column-by-column path.  python tools/marginals_time.py [poses landmarks]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openslam_g2o_amd import lm, synthetic as S

P, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 1000000)
pr = S.make_ba_problem(P, L)
s, g = lm.setup_device_ba(pr)
g.linearize()
s.buildSystem()
s.setLambda(1.0, True)
idx = np.arange(pr["nP"], dtype=np.int32)
t0 = time.perf_counter(); M = s.computeMarginals(idx, idx); t1 = time.perf_counter()
t2 = time.perf_counter(); M = s.computeMarginals(idx, idx); t3 = time.perf_counter()
s.setOption("marginals_recursion", 0)
sel = idx[::max(1, len(idx) // 20)][:20]
t4 = time.perf_counter(); M0 = s.computeMarginals(sel, sel); t5 = time.perf_counter()
err = float(np.abs(M[sel] - M0).max() / np.abs(M0).max())
print(json.dumps({"poses": int(pr["nP"]), "blocks": int(len(idx)), "recursion_first_s": t1 - t0, "recursion_s": t3 - t2,
                  "column_path_s_per_block": (t5 - t4) / len(sel), "column_path_extrapolated_s": (t5 - t4) / len(sel) * len(idx),
                  "max_rel_diff_on_sample": err}))
