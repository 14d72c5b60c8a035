"""Generates the committed golden fixtures under tests/golden/ (run HERE, where
/root/reference and oracle/_ref exist):

    python tests/golden/make_golden.py

Fixtures are data only -- parsed graph inputs and expected outputs:
  * manhattan3500.npz : the parse result of data/2d/manhattan3500/manhattanOlson3500.g2o
    (vertex estimates, edge endpoints, measurements, information upper triangles) and the
    reference CSparse path's answers for the first Gauss-Newton linear system
    (x, lnz, chi2), with and without LM damping, plus a 5-iteration GN chi2 trajectory.
    The expected x comes from the REFERENCE's own compiled code
    (cs_amd block ordering + csparse_extension::cs_cholsolsymb, oracle/_ref).
  * sphere2200.npz : the same for data/3d/sphere/sphere_bignoise_vertex3.g2o (config 2; the
    bundled file has 2 200 vertices / 8 647 EDGE_SE3:QUAT): first-iteration x (plain and damped)
    from the reference CSparse path, lnz, and the chi2 of three damped iterations.
  * ba_small.npz : expected outputs for the deterministic synthetic 20-pose / 200-point BA
    problem (inputs are regenerated from the seed): x from the reference CSparse path on
    the Schur-reduced system and from a dense solve of the full system.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from openslam_g2o_amd import g2o_io, synthetic as S  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def manhattan():
    g = g2o_io.read_g2o("/root/reference/data/2d/manhattan3500/manhattanOlson3500.g2o")
    nv = len(g["ids"])
    h, nP = g2o_io.hessian_index(nv, [0])       # the repo fixes the lowest id
    om = g["info"].transpose(0, 2, 1).reshape(-1, 9)
    est = g["estimates"].copy()
    out = dict(estimates=g["estimates"], vi=g["vi"], vj=g["vj"], meas=g["meas"],
               info_upper=np.stack([g["info"][:, i, j] for i in range(3) for j in range(i, 3)], axis=1))
    chi_traj = []
    for it in range(5):
        J0, J1, err = O.se2_edges(est, g["vi"], g["vj"], g["meas"])
        s = O.OracleSolver(3, 2, nP, 0, schur=False)
        k = s.add_edge_set(3, h[g["vi"]], h[g["vj"]])
        s.set_dims(k, 3, 3)
        s.build_structure()
        s.set_edge_data(k, J0, J1, om, err)
        s.build_system()
        chi_traj.append(s.chi2())
        cp, row = s.pattern("pp")
        val = s.values("Hpp")
        ok, x, lnz, P = O.ref_solve_blocks(nP, 3, cp, row, val, s.b())
        assert ok
        if it == 0:
            out.update(x_gn0=x, lnz_block_amd=lnz, block_perm=P, b0=s.b(), nnzb=len(row))
            lam = 1e-5 * s.max_diagonal()           # computeLambdaInit, tau = 1e-5
            s.set_lambda(lam, True)
            ok2, x2, _, _ = O.ref_solve_blocks(nP, 3, cp, row, s.values("Hpp"), s.b())
            s.restore_diagonal()
            assert ok2
            out.update(lambda0=lam, x_lm0=x2)
        est = O.se2_oplus(est, h, x)
    out["chi2_gn"] = np.asarray(chi_traj)
    np.savez_compressed(os.path.join(OUT, "manhattan3500.npz"), **out)
    print("manhattan: nP", nP, "nnzb", out["nnzb"], "lnz", out["lnz_block_amd"], "chi2", chi_traj)


def ba_small():
    pr = S.make_ba_problem(20, 200)
    Jp, Jc, e = S.ba_linearize(pr)
    om = S.ba_omega(pr)
    lam = 1.0
    s = O.OracleSolver(6, 3, pr["nP"], pr["nL"], True)
    k = s.add_edge_set(2, pr["v0"], pr["v1"])
    s.set_dims(k, 3, 6)
    s.build_structure()
    s.set_edge_data(k, Jp, Jc, om, e)
    s.build_system()
    chi2 = s.chi2()
    b = s.b()
    s.set_lambda(lam, True)
    assert s.solve()          # fills Hschur / bschur
    cp, row = s.pattern("hs")
    ok, xp, lnz, _ = O.ref_solve_blocks(pr["nP"], 6, cp, row, s.values("Hschur"), s.bschur())
    assert ok
    H = s.dense_full()
    x_dense = np.linalg.solve(H, b)
    x_schur = s.x()
    assert np.abs(x_schur[:6 * pr["nP"]] - xp).max() < 1e-9 * np.abs(xp).max()
    assert np.abs(x_schur - x_dense).max() < 1e-9 * np.abs(x_dense).max()
    np.savez_compressed(os.path.join(OUT, "ba_small.npz"), lam=lam, chi2=chi2, b=b, x_dense=x_dense, xp_ref_csparse=xp,
                        Hschur=s.values("Hschur"), bschur=s.bschur(), hs_colptr=cp, hs_row=row, lnz=lnz,
                        meas_checksum=float(np.sum(pr["meas"])), pts_checksum=float(np.sum(pr["pts"])))
    print("ba_small: chi2", chi2, "lnz", lnz)


def sphere():
    """Config 2 input: data/3d/sphere/sphere_bignoise_vertex3.g2o (2 200 VERTEX_SE3:QUAT, 8 647
    EDGE_SE3:QUAT), vertex 0 fixed, BlockSolver_6_3 semantics without Schur."""
    g = g2o_io.read_g2o("/root/reference/data/3d/sphere/sphere_bignoise_vertex3.g2o")
    nv = len(g["ids"])
    h, nP = g2o_io.hessian_index(nv, [0])
    om = g["info"].transpose(0, 2, 1).reshape(-1, 36)
    poses = O.se3_from_qt(g["estimates"], normalize=False)          # VertexSE3::read: fromVectorQT
    Z = O.se3_from_qt(g["meas"], normalize=True)                    # EdgeSE3::read normalises the quaternion
    out = dict(estimates=g["estimates"], vi=g["vi"], vj=g["vj"], meas=g["meas"],
               info_upper=np.stack([g["info"][:, i, j] for i in range(6) for j in range(i, 6)], axis=1))
    chi_traj = []
    lam = None
    for it in range(3):
        J0, J1, err = O.se3_edges(poses, g["vi"], g["vj"], Z)
        s = O.OracleSolver(6, 3, nP, 0, schur=False)
        k = s.add_edge_set(6, h[g["vi"]], h[g["vj"]])
        s.set_dims(k, 6, 6)
        s.build_structure()
        s.set_edge_data(k, J0, J1, om, err)
        s.build_system()
        chi_traj.append(s.chi2())
        cp, row = s.pattern("pp")
        if it == 0:
            ok, x, lnz, P = O.ref_solve_blocks(nP, 6, cp, row, s.values("Hpp"), s.b())
            assert ok
            lam = 1e-5 * s.max_diagonal()
            out.update(x_gn0=x, lnz_block_amd=lnz, block_perm=P, b0=s.b(), nnzb=len(row), lambda0=lam)
        # damped steps (plain GN diverges on this very noisy initial guess): x = (H + lam I)^-1 b
        s.set_lambda(lam, True)
        ok2, x2, _, _ = O.ref_solve_blocks(nP, 6, cp, row, s.values("Hpp"), s.b())
        s.restore_diagonal()
        assert ok2
        if it == 0:
            out["x_lm0"] = x2
        poses = O.se3_oplus(poses, h, x2)
    out["chi2_lm"] = np.asarray(chi_traj)
    np.savez_compressed(os.path.join(OUT, "sphere2200.npz"), **out)
    print("sphere: nP", nP, "nnzb", out["nnzb"], "lnz", out["lnz_block_amd"], "chi2", chi_traj)


def _main():
    O.build()
    assert O.ref() is not None, "oracle/_ref missing: run `make -C oracle` where /root/reference exists"
    if "lnz" in sys.argv[1:]:
        return lnz_amd()
    manhattan()
    ba_small()
    sphere()
    lnz_amd()


def lnz_amd():
    """nnz(L) of the reduced BA system under the REFERENCE's ordering (cs_amd on the block pattern, scalar blow-up,
    reference symbolic analysis: linear_solver_csparse.h:246-308 through oracle/_ref) for the synthetic window graphs the
    GPU tests and the bench use.  The reduced pattern of SURVEY.md 8d's generator: pose blocks i, j coupled iff |i - j| <= 4
    (5 consecutive observing poses per landmark), poses 0 and 1 fixed -> nb = P - 2 block columns."""
    import json
    R = O.ref()
    out = {}
    for P in (300, 2000, 20000, 50000, 100000):
        nb = P - 2
        cp = [0]
        ri = []
        for c in range(nb):
            for r in range(max(0, c - 4), c + 1):
                ri.append(r)
            cp.append(len(ri))
        cp, ri = np.asarray(cp, np.int32), np.asarray(ri, np.int32)
        perm = np.zeros(nb, np.int32)
        assert R.ref_block_amd(nb, O._ip(cp), O._ip(ri), O._ip(perm))
        # scalar pattern (values irrelevant) + reference symbolic with the expanded permutation
        val = np.zeros(len(ri) * 36)
        Ap, Ai, Ax = O.scalar_ccs(nb, 6, cp, ri, val)
        sperm = (perm[:, None] * 6 + np.arange(6, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)
        sym = R.ref_symbolic(nb * 6, O._ip(Ap), O._ip(Ai), O._ip(sperm))
        lnz = R.ref_lnz(sym)
        R.ref_free(sym)
        out[str(P)] = dict(block_columns=nb, upper_blocks=int(len(ri)), lnz_block_amd=float(lnz))
        print("lnz_amd: P", P, "nb", nb, "lnz", lnz)
    with open(os.path.join(OUT, "lnz_amd.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    _main()
