"""Undamped (Gauss-Newton) solve of the synthetic BA graph at growing sizes: does the factorisation of the reduced system break down
(d <= 0) where the CPU oracle's does not?  python tools/probe/gn_lambda0.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openslam_g2o_amd import lm, synthetic as S
from tests.helpers import oracle_ba

for P, L in ((2000, 20000), (20000, 200000), (100000, 1000000)):
    pr = S.make_ba_problem(P, L)
    s, g = lm.setup_device_ba(pr)
    g.linearize()
    s.buildSystem()
    md = s.maxDiagonal()
    for lam in (0.0, 1e-14 * md, 1e-12 * md, 1e-10 * md):
        s.setLambda(lam, True)
        ok = s.solve()
        x = s.x() if ok else None
        s.restoreDiagonal()
        line = "P %d lambda %.3g gpu ok %s |x|inf %s" % (P, lam, ok, None if x is None else float(np.abs(x).max()))
        if P <= 20000:
            Jp, Jc, err = S.ba_linearize(pr)
            pr.update(Jp=Jp, Jc=Jc, err=err, omega=S.ba_omega(pr))
            o = oracle_ba(pr)
            o.build_system()
            o.set_lambda(lam, True)
            oko = o.solve()
            xo = o.x()
            line += " | oracle ok %s |x|inf %.6g" % (oko, float(np.abs(xo).max()))
            if ok and oko:
                line += " rel diff %.3g" % (float(np.abs(x - xo).max()) / float(np.abs(xo).max()))
        print(line, flush=True)
