"""GPU: the one-time set-up runs its big host loops on G2OHIP_HOST_THREADS threads (Schur tile entry lists, contributor
lists, the per-observation copies of the BA front end: common.h host_parallel_for / host_parallel_chunks).  Every chunk
writes its own output and the chunks are concatenated in order, so the tables -- and with them every sum on the device --
must not depend on the number of threads: one solve with 1 thread and one with 8, in two processes (the count is read once
per process), bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from openslam_g2o_amd import capi, lm, synthetic as S
pr = S.make_ba_problem(6000, 60000, seed=7)        # 300 000 observations: several chunks of every threaded loop, > 200 tiles
s, g = lm.setup_device_ba(pr, huber_delta=1.0)
g.linearize()
s.buildSystem()
s.setLambda(10.0, True)
assert s.solve()
s.restoreDiagonal()
x = s.x()
hs = s.values(capi.HSCHUR)
np.savez(sys.argv[1], x=x, b=s.b(), chi2=np.array([s.chi2()]), hs=np.asarray(hs))
"""


def _run(tmp_path, threads, options=""):
    out = str(tmp_path / ("t%d%s.npz" % (threads, "o" if options else "")))
    env = dict(os.environ, G2OHIP_HOST_THREADS=str(threads))
    if options:
        env["G2OHIP_OPTIONS"] = options
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT, out], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def test_setup_tables_do_not_depend_on_the_number_of_host_threads(tmp_path):
    a, b = _run(tmp_path, 1), _run(tmp_path, 8)
    c = _run(tmp_path, 8, "setup_overlap=0")     # (the symbolic analysis behind the Schur tiles' set-up instead of next to it)
    for k in ("x", "b", "chi2", "hs"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], c[k]), k
    assert np.isfinite(a["x"]).all() and np.abs(a["x"]).max() > 0
