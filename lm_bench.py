#!/usr/bin/env python
"""Secondary measurement (not the headline metric): whole Levenberg-Marquardt iterations of BASELINE.json's
config 5 -- 100k-pose / 1M-landmark BA, Huber kernel (delta = 1), 5 % outliers, tau = 1e-5, <= 10 trials --
with estimates, errors and Jacobians resident on the device.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=100000)
    ap.add_argument("--landmarks", type=int, default=1000000)
    ap.add_argument("--iterations", type=int, default=10)
    args = ap.parse_args()
    from openslam_g2o_amd import lm, synthetic as S
    prob = S.make_ba_problem(args.poses, args.landmarks, outlier_frac=0.05)
    s, g = lm.setup_device_ba(prob, huber_delta=1.0)
    g.compute_active_errors()
    chi0 = g.chi2()
    s.sync()
    t0 = time.perf_counter()
    n, chis, lams, trials = lm.optimize(g, s, args.iterations, "lm")
    s.sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "config 5: %d poses / %d landmarks / %d observations, Huber delta=1, 5%% outliers, LM tau=1e-5" % (
        args.poses, args.landmarks, prob["E"]), "iterations": n, "lm_trials": trials, "ms_per_lm_iteration": 1e3 * dt / max(n, 1),
        "ms_per_lm_trial": 1e3 * dt / max(sum(trials), 1), "chi2_initial": chi0, "chi2": chis, "lambda": lams}))


if __name__ == "__main__":
    main()
