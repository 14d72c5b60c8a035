"""CPU-only host logic: the C-ABI library loads and exports every symbol include/g2ohip.h
declares, fails loudly without a GPU, and the host-side helpers behave."""
import ctypes
import os
import re

import numpy as np
import pytest

from openslam_g2o_amd import capi, g2o_io, synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "g2ohip.h")).read()
    declared = sorted(set(re.findall(r"\b(g2ohip_\w+)\s*\(", hdr)))
    assert len(declared) >= 40
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "missing export %s" % name
    assert sorted(capi.EXPORTS) == declared


def test_no_cpu_fallback_without_gpu():
    L = capi.load()
    if L.g2ohip_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.G2oHipError):
        capi.HipBlockSolver(6, 3, 0)
    with pytest.raises(capi.G2oHipError):
        capi.HipLinearSolver(6, 0)


def test_generator_is_deterministic_and_shaped():
    a = S.make_ba_problem(30, 300)
    b = S.make_ba_problem(30, 300)
    for k in ("cams", "pts", "meas", "v0", "v1"):
        assert np.array_equal(a[k], b[k])
    assert a["E"] == 1500 and a["nP"] == 28 and (a["v1"] >= -1).all() and a["v0"].min() == a["nP"]
    # every landmark is seen by 5 distinct poses
    assert (np.diff(a["cam_idx"].reshape(-1, 5), axis=1) == 1).all()
    Jp, Jc, err = S.ba_linearize(a)
    assert Jp.shape == (1500, 6) and Jc.shape == (1500, 12) and err.shape == (1500, 2)
    # numeric check of the analytic Jacobians (the reference's own fallback: central differences)
    x = np.zeros(6 * a["nP"] + 3 * a["nL"])
    k = 7
    h = 1e-6
    lm = a["pt_idx"][k]
    for c in range(3):
        xx = x.copy()
        xx[6 * a["nP"] + 3 * lm + c] = h
        ep = S.ba_linearize(S.ba_oplus(a, xx), jac=False)[k]
        em = S.ba_linearize(S.ba_oplus(a, -xx), jac=False)[k]
        assert np.allclose((ep - em) / (2 * h), Jp[k].reshape(3, 2)[c], rtol=1e-5, atol=1e-5)
    cam = a["cam_hidx"][a["cam_idx"][k]]
    if cam >= 0:
        for c in range(6):
            xx = x.copy()
            xx[6 * cam + c] = h
            ep = S.ba_linearize(S.ba_oplus(a, xx), jac=False)[k]
            em = S.ba_linearize(S.ba_oplus(a, -xx), jac=False)[k]
            assert np.allclose((ep - em) / (2 * h), Jc[k].reshape(6, 2)[c], rtol=1e-4, atol=1e-3)


def test_grid_generator_is_deterministic_ragged_and_consistent_with_the_oracle():
    """synthetic.make_ba_grid (visibility by distance, round 6): deterministic, point-major edges with cameras ascending inside a
    list, ragged observation lists, every camera coupled to many others; the oracle solves the 100-camera instance and the Schur
    path agrees with the dense full-system solve (the identity of SURVEY.md 8c, golden vector 2)."""
    from oracle import oracle as O
    a, b = S.make_ba_grid(100), S.make_ba_grid(100)
    for k in ("cams", "pts", "meas", "cam_idx", "pt_idx", "v0", "v1"):
        assert np.array_equal(a[k], b[k])
    assert a["P"] == 100 and a["L"] == 1000 and a["nP"] == 98
    K = np.bincount(a["pt_idx"], minlength=a["L"])
    assert K.min() >= 2 and K.max() >= 10 and len(np.unique(K)) > 3          # ragged
    assert (np.diff(a["pt_idx"]) >= 0).all()                                   # point-major
    same = np.diff(a["pt_idx"]) == 0
    assert (np.diff(a["cam_idx"])[same] > 0).all()                             # cameras ascending inside a list
    Jp, Jc, err = S.ba_linearize(a)
    o = O.OracleSolver(6, 3, a["nP"], a["nL"], True)
    k = o.add_edge_set(2, a["v0"], a["v1"])
    o.set_dims(k, 3, 6)
    o.build_structure()
    o.set_edge_data(k, Jp, Jc, S.ba_omega(a), err)
    o.build_system()
    o.set_lambda(10.0, True)
    assert o.solve()
    cp, ri = o.pattern("hs")
    deg = np.diff(cp)
    assert deg.max() >= 15
    H = o.dense_full()
    x = np.linalg.solve(H, o.b())
    assert np.abs(o.x() - x).max() <= 1e-9 * np.abs(x).max()


def test_g2o_reader(tmp_path):
    p = tmp_path / "t.g2o"
    p.write_text("VERTEX_SE2 3 1 2 0.5\nVERTEX_SE2 1 0 0 0\nEDGE_SE2 1 3 1 2 0.5 10 1 2 20 3 30\nFIX 1\n")
    g = g2o_io.read_g2o(str(p))
    assert g["kind"] == "se2" and list(g["ids"]) == [1, 3] and g["fixed"] == [0]
    assert list(g["vi"]) == [0] and list(g["vj"]) == [1]
    assert np.allclose(g["info"][0], [[10, 1, 2], [1, 20, 3], [2, 3, 30]])
    h, n = g2o_io.hessian_index(2, g["fixed"])
    assert list(h) == [-1, 0] and n == 1
