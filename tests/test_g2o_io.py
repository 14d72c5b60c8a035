"""`.g2o` bundle-adjustment tags: write a synthetic problem with the reference's text format
(VERTEX_SE3:EXPMAP holds cam->world, the vertex the inverse), read it back, and get the same linearised system."""
import numpy as np

from openslam_g2o_amd import g2o_io, synthetic as S
from tests.helpers import ba_case, oracle_ba


def test_ba_tags_round_trip(tmp_path):
    pr = ba_case(12, 80)
    path = str(tmp_path / "ba.g2o")
    g2o_io.write_g2o_ba(path, pr)
    text = open(path).read().split("\n")
    assert text[0].startswith("PARAMS_CAMERAPARAMETERS 0 1000 320 240")
    assert sum(l.startswith("VERTEX_SE3:EXPMAP") for l in text) == 12 and sum(l.startswith("VERTEX_XYZ") for l in text) == 80
    assert any(l == "FIX 0 1" for l in text)
    rd = g2o_io.read_g2o_ba(path)
    for k in ("P", "L", "E", "nP", "nL"):
        assert rd[k] == pr[k]
    for k in ("cam_idx", "pt_idx", "cam_hidx", "v0", "v1"):
        assert np.array_equal(rd[k], pr[k])
    assert np.abs(rd["cams"] - pr["cams"]).max() < 1e-14 and np.abs(rd["pts"] - pr["pts"]).max() == 0
    assert np.abs(rd["meas"] - pr["meas"]).max() == 0 and np.abs(rd["omega"] - pr["omega"]).max() == 0
    # same linearised system and the same first Gauss-Newton step through the oracle
    Jp, Jc, err = S.ba_linearize(rd)
    rd.update(Jp=Jp, Jc=Jc, err=err)
    a, b = oracle_ba(pr), oracle_ba(rd)
    a.build_system(); b.build_system()
    assert abs(a.chi2() - b.chi2()) <= 1e-12 * a.chi2()
    a.set_lambda(1.0, True); b.set_lambda(1.0, True)
    assert a.solve() and b.solve()
    assert np.abs(a.x() - b.x()).max() <= 1e-9 * np.abs(a.x()).max()
