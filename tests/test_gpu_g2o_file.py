"""A bundle-adjustment `.g2o` FILE through the HIP path: the problem is written with the reference's text tags
(PARAMS_CAMERAPARAMETERS, VERTEX_SE3:EXPMAP -- the file holds camera -> world, the vertex the inverse,
types_six_dof_expmap.cpp:88-103 --, VERTEX_XYZ, EDGE_PROJECT_XYZ2UV:EXPMAP, FIX; optimizable_graph.cpp:356-622), read back, and
optimised on the device; the trajectory equals the one of the CPU oracle on the problem the file was written from."""
import numpy as np
import pytest

from openslam_g2o_amd import g2o_io, lm, synthetic as S
from tests.helpers import ba_case, relerr

pytestmark = pytest.mark.gpu


def _case(kind):
    if kind == "band":
        return ba_case(20, 200)
    pr = S.make_ba_loops(60, 260, laps=3, hubs=1)
    Jp, Jc, err = S.ba_linearize(pr)
    pr.update(Jp=Jp, Jc=Jc, err=err, omega=S.ba_omega(pr))
    return pr


@pytest.mark.parametrize("kind", ["band", "loops"])
def test_ba_file_to_device_levenberg_marquardt(tmp_path, kind):
    from tests.test_gpu_lm import OracleBAGraph, OracleSolverAdapter
    pr = _case(kind)
    path = str(tmp_path / "ba.g2o")
    g2o_io.write_g2o_ba(path, pr)
    text = open(path).read()
    for tag in ("PARAMS_CAMERAPARAMETERS", "VERTEX_SE3:EXPMAP", "VERTEX_XYZ", "EDGE_PROJECT_XYZ2UV:EXPMAP", "FIX"):
        assert tag in text
    rd = g2o_io.read_g2o_ba(path)
    assert rd["E"] == pr["E"] and rd["nP"] == pr["nP"] and rd["nL"] == pr["nL"] and np.array_equal(rd["cam_hidx"], pr["cam_hidx"])
    # device: everything from the FILE
    s, g = lm.setup_device_ba(rd, huber_delta=1.0)
    g.compute_active_errors()
    chi0 = g.chi2()
    n, chis, lams, trials = lm.optimize(g, s, 6, "lm")
    cams, pts = s.baGetEstimates()
    # oracle: the problem the file was written from
    go = OracleBAGraph(pr, huber=1.0)
    go.linearize()
    chi0_o = go.chi2()
    n_o, chis_o, lams_o, trials_o = lm.optimize(go, OracleSolverAdapter(go.o), 6, "lm")
    assert abs(chi0 - chi0_o) <= 1e-9 * chi0_o
    assert n == n_o and trials == trials_o
    assert np.allclose(chis, chis_o, rtol=1e-6, atol=0) and np.allclose(lams, lams_o, rtol=1e-6, atol=0)
    assert chis[-1] < chi0
    assert relerr(cams, go.pr["cams"]) < 1e-6 and relerr(pts, go.pr["pts"]) < 1e-6
    # and the written-back file carries the optimised estimates in the file convention
    rd2 = dict(rd)
    rd2["cams"], rd2["pts"] = cams, pts
    out = str(tmp_path / "out.g2o")
    g2o_io.write_g2o_ba(out, rd2)
    back = g2o_io.read_g2o_ba(out)
    assert np.abs(back["cams"] - cams).max() < 1e-12 and np.abs(back["pts"] - pts).max() == 0
