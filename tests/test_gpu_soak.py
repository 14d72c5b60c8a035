"""Soak of the in-launch hand-offs (dependency counters between tree levels, the band chains' signals to their parents) with
TWO solver handles running concurrently on two streams of one GPU -- the shared-GPU set-up of the multi-rank tests, where
another queue's kernels compete for the CUs while parents wait for their children: every solve must reproduce its first
solution bit for bit and no launch may fall back to one launch per level (g2ohip_stats.dependencyFallbacks == 0)."""
import threading

import numpy as np
import pytest

from tests.helpers import ba_case, hip_ba

pytestmark = pytest.mark.gpu


def _worker(pr, n, out, idx, graph):
    try:
        s = hip_ba(pr, options={"use_graph": graph})
        s.buildSystem()
        x0, bad = None, 0
        for _ in range(n):
            s.setLambda(7.0, True)
            ok = s.solve()
            s.restoreDiagonal()
            x = s.x()
            if x0 is None:
                x0 = x
            if not ok or not np.array_equal(x, x0):
                bad += 1
        st = s.stats()
        out[idx] = (bad, st["dependencyFallbacks"], st["bandChains"], st["numLevels"], x0)
    except Exception as e:      # noqa: BLE001
        out[idx] = e


@pytest.mark.parametrize("graph", [0, 1])
def test_two_solvers_on_two_streams_1000_solves_each(graph):
    prs = [ba_case(6000, 60000), ba_case(9000, 90000, seed=7)]       # different sizes: the two queues drift against each other
    out = [None, None]
    th = [threading.Thread(target=_worker, args=(prs[i], 1000, out, i, graph)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        assert not isinstance(out[i], Exception), out[i]
        bad, fallbacks, chains, levels, x0 = out[i]
        assert chains > 0 and levels >= 6                 # band chains signalling parents inside a dependency-driven launch
        assert bad == 0 and fallbacks == 0, (i, bad, fallbacks)
    # and alone, the same solver gives the same bits as it did next to the other one
    s = hip_ba(prs[0], options={"use_graph": graph})
    s.buildSystem()
    s.setLambda(7.0, True)
    assert s.solve()
    assert np.array_equal(s.x(), out[0][4])
