"""Subtree-distributed solve (openslam_g2o_amd/distributed.py mode "subtree") on real hardware:
world_size 2 and 3 processes sharing cuda:0, exchange through HostStagedComm (gloo) because RCCL
refuses two ranks on one device.  Everything else is the production N>1 path: pose partition from
the elimination-task tree, landmarks dealt by pose owner, boundary-block exchange, per-rank lambda
mask, own-subtree factorisation, exchange of the subtree roots, shared top, masked x_p all-reduce,
sharded back-substitution.  Checked against the unsharded CPU oracle."""
import os
import socket

import numpy as np
import pytest

from tests.helpers import ba_case, dx_tolerance, oracle_ba, relerr

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, P, L, lam, fused, x_exchange, out_dir, options=""):
    if options:
        os.environ["G2OHIP_OPTIONS"] = options
    import torch
    import torch.distributed as dist
    from openslam_g2o_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        pr = _case(P, L)
        s = D.ShardedBlockSolver(6, 3, rank=rank, world=world, comm=D.HostStagedComm(world), mode="subtree", x_exchange=x_exchange)
        info = s.setup_ba(pr, torch_device=dev, fused=fused)
        chis, xs = [], []
        for it in range(2):               # twice: the phased factorisation must be repeatable
            s.buildSystem()
            chis.append(s.chi2())
            s.setLambda(lam, True)
            ok = s.solve()
            s.restoreDiagonal()
            xs.append(s.local.x())
        assert np.array_equal(xs[0], xs[1])
        owned = np.bincount(s.pose_owner + 1, minlength=world + 1)
        xloc = s.x_poses()
        valid = (s.pose_owner == rank) | (s.pose_owner < 0)
        valid[s.halo] = True            # own + shared + halo poses are valid without any gather
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), ok=ok, xp=s.gather_x_poses(), xl=s.x_landmarks_local(),
                 xloc=xloc, valid=valid, halo=len(s.halo), bposes=len(s.bposes),
                 lm_index=s.lm_index, chi2=chis[0], owned=owned, boundary=len(s.boundary), nnzb=s.nnzb_reduced,
                 volume=s.exchange_volume(), E_local=info["E_local"])
    finally:
        dist.destroy_process_group()


def _case(P, L):
    """L < 0: the graph with loop closures, ragged lists and a hub point (synthetic.make_ba_loops) instead of the band; L == 0: the
    grid graph with visibility by distance (synthetic.make_ba_grid, P cameras)."""
    if L > 0:
        return ba_case(P, L)
    from openslam_g2o_amd import synthetic as S
    pr = S.make_ba_grid(P) if L == 0 else S.make_ba_loops(P, -L, laps=4, hubs=1)
    Jp, Jc, err = S.ba_linearize(pr)
    pr.update(Jp=Jp, Jc=Jc, err=err, omega=S.ba_omega(pr))
    return pr


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,fused,x_exchange", [(2, False, "full"), (3, True, "halo"), (2, True, "halo"), (4, True, "halo")])
def test_subtree_distributed_solve_matches_oracle(tmp_path, world, fused, x_exchange):
    import torch.multiprocessing as mp
    P, L, lam = 700, 6000, 30.0
    mp.spawn(_worker, args=(world, _free_port(), P, L, lam, fused, x_exchange, str(tmp_path)), nprocs=world, join=True)
    pr = ba_case(P, L)
    o = oracle_ba(pr)
    o.build_system()
    chi2 = o.chi2()
    o.set_lambda(lam, True)
    assert o.solve()
    x = o.x()
    tol = dx_tolerance(o)[0]          # SURVEY 8d's 1e-8, or the cond-scaled bound where the reduced system is ill-conditioned
    nP = pr["nP"]
    xp, xl = x[:6 * nP], x[6 * nP:].reshape(-1, 3)
    seen = np.zeros(pr["nL"], int)
    edges = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert bool(z["ok"])
        assert abs(float(z["chi2"]) - chi2) <= 1e-9 * chi2
        assert np.abs(z["xp"] - xp).max() <= tol * np.abs(x).max()          # gathered x_p
        v = np.repeat(z["valid"], 6) if x_exchange == "halo" else np.ones(6 * nP, bool)
        assert np.abs(z["xloc"][v] - xp[v]).max() <= tol * np.abs(x).max()  # own + shared + halo poses without the gather
        assert int(z["halo"]) < 0.1 * nP and int(z["bposes"]) < 0.2 * nP
        idx = z["lm_index"]
        seen[idx] += 1
        assert np.abs(z["xl"].reshape(-1, 3) - xl[idx]).max() <= tol * np.abs(x).max()   # x_l sharded by owner
        owned = z["owned"]
        assert owned[1:].min() > 0.6 * nP / world and owned[0] < 0.2 * nP      # balanced subtrees, small shared top
        assert int(z["boundary"]) < 0.25 * int(z["nnzb"])                     # most Schur blocks never leave their rank
        edges += int(z["E_local"])
    assert (seen == 1).all() and edges == pr["E"]


@pytest.mark.parametrize("world,P,L", [(2, 420, -1600), (3, 420, -1600), (2, 400, 0), (4, 400, 0)])
def test_subtree_distributed_solve_on_a_graph_with_loop_closures(tmp_path, world, P, L):
    """The same sharded solve on graphs that are not a band: the loop-closure graph (the elimination tree has a dense top, most of it
    shared, the subtrees are uneven, a hub point couples a third of the poses) and the grid graph with visibility by distance (a
    two-dimensional mesh: every separator is shared by several ranks' subtrees).  Only correctness is asserted."""
    import torch.multiprocessing as mp
    lam = 30.0
    mp.spawn(_worker, args=(world, _free_port(), P, L, lam, True, "halo", str(tmp_path)), nprocs=world, join=True)
    pr = _case(P, L)
    o = oracle_ba(pr)
    o.build_system()
    chi2 = o.chi2()
    o.set_lambda(lam, True)
    assert o.solve()
    x = o.x()
    tol = dx_tolerance(o)[0]          # SURVEY 8d's 1e-8, or the cond-scaled bound where the reduced system is ill-conditioned
    nP = pr["nP"]
    xp, xl = x[:6 * nP], x[6 * nP:].reshape(-1, 3)
    seen = np.zeros(pr["nL"], int)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert bool(z["ok"])
        assert abs(float(z["chi2"]) - chi2) <= 1e-9 * chi2
        assert np.abs(z["xp"] - xp).max() <= tol * np.abs(x).max()
        v = np.repeat(z["valid"], 6)
        assert np.abs(z["xloc"][v] - xp[v]).max() <= tol * np.abs(x).max()
        idx = z["lm_index"]
        seen[idx] += 1
        assert np.abs(z["xl"].reshape(-1, 3) - xl[idx]).max() <= tol * np.abs(x).max()
    assert (seen == 1).all()


def _pcg_worker(rank, world, port, P, L, lam, out_dir):
    import torch
    import torch.distributed as dist
    from openslam_g2o_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        pr = ba_case(P, L)
        s = D.ShardedBlockSolver(6, 3, rank=rank, world=world, comm=D.HostStagedComm(world), mode="pcg")
        s.pcg_tolerance, s.pcg_max_iterations = 1e-20, 8000
        s.setup_ba(pr, torch_device=dev, fused=True)
        s.buildSystem()
        chi = s.chi2()
        s.setLambda(lam, True)
        ok = s.solve()
        s.restoreDiagonal()
        np.savez(os.path.join(out_dir, "p%d.npz" % rank), ok=ok, xp=s.x_poses(), xl=s.x_landmarks_local(), lm_index=s.lm_index,
                 chi2=chi, iters=s.pcg_iterations)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_matrix_free_pcg_matches_oracle(tmp_path, world):
    """mode "pcg": landmarks sharded by range, the reduced system never formed, every rank applies its summand of
    Hschur*d and one all-reduce per iteration completes it (exchange staged through gloo: the ranks share one GPU)."""
    import torch.multiprocessing as mp
    P, L, lam = 150, 1800, 20.0
    mp.spawn(_pcg_worker, args=(world, _free_port(), P, L, lam, str(tmp_path)), nprocs=world, join=True)
    pr = ba_case(P, L)
    o = oracle_ba(pr)
    o.build_system()
    chi2 = o.chi2()
    o.set_lambda(lam, True)
    assert o.solve()
    x = o.x()
    tol = dx_tolerance(o)[0]          # SURVEY 8d's 1e-8, or the cond-scaled bound where the reduced system is ill-conditioned
    nP = pr["nP"]
    xp, xl = x[:6 * nP], x[6 * nP:].reshape(-1, 3)
    its = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "p%d.npz" % r))
        assert bool(z["ok"]) and abs(float(z["chi2"]) - chi2) <= 1e-9 * chi2
        assert np.abs(z["xp"] - xp).max() <= 1e-6 * np.abs(xp).max()             # x_p replicated
        assert np.abs(z["xl"].reshape(-1, 3) - xl[z["lm_index"]]).max() <= 1e-6 * np.abs(xl).max()
        its.append(int(z["iters"]))
    assert len(set(its)) == 1 and its[0] > 0                                      # every rank took the same decisions


def _lm_worker(rank, world, port, P, L, huber, outliers, n_it, out_dir, stall_rank=-1, options="", lib_comm="host"):
    if options:
        os.environ["G2OHIP_OPTIONS"] = options
    if rank == stall_rank:       # this rank's dependency-driven launches give up at once (the safety net under test)
        os.environ["G2OHIP_OPTIONS"] = "dep_spin_limit=0"
    import torch
    import torch.distributed as dist
    from openslam_g2o_amd import capi, distributed as D, lm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        pr = ba_case(P, L, outlier_frac=outliers)
        s = D.ShardedBlockSolver(6, 3, rank=rank, world=world, comm=D.HostStagedComm(world), mode="subtree")
        s.setup_ba(pr, torch_device=dev, fused=True)
        assert s.attach_library_comm(lib_comm)    # collectives inside libg2ohip (host callback: the ranks share one GPU)
        if huber > 0:
            s.setRobustKernel(capi.KERNEL_HUBER, huber)
        g = D.ShardedBAGraph(s)
        done, chis, lams, trials = lm.optimize(g, s, n_it, "lm")
        cams, pts = s.local.baGetEstimates()
        np.savez(os.path.join(out_dir, "lm%d.npz" % rank), done=done, chis=chis, lams=lams, trials=trials, cams=cams, pts=pts,
                 lm_index=s.lm_index, pose_owner=s.pose_owner, halo=s.halo, fallbacks=s.local.stats()["dependencyFallbacks"],
                 collectives=s.local.stats()["shardedCollectives"])
    finally:
        dist.destroy_process_group()


def test_a_stall_on_one_rank_repeats_the_solve_on_all_ranks(tmp_path):
    """A dependency-driven launch that gives up waiting on ONE rank (spin limit 0 there) must not look like "not positive
    definite" to the LM loop: the status word all ranks share tells the two apart (G2OHIP_REPEAT), every rank runs the
    solve again, the stalled rank with one launch per level -- same trials and chi2 as the single-rank run."""
    import torch.multiprocessing as mp
    from openslam_g2o_amd import lm
    world, P, L, n_it = 2, 400, 3600, 4
    mp.spawn(_lm_worker, args=(world, _free_port(), P, L, 0.0, 0.0, n_it, str(tmp_path), 1), nprocs=world, join=True)
    pr = ba_case(P, L)
    s1, g1 = lm.setup_device_ba(pr)
    done1, chis1, lams1, trials1 = lm.optimize(g1, s1, n_it, "lm")
    z = [np.load(os.path.join(str(tmp_path), "lm%d.npz" % r)) for r in range(world)]
    assert int(z[1]["fallbacks"]) >= 1 and int(z[0]["fallbacks"]) == 0
    for r in range(world):
        assert int(z[r]["done"]) == done1 and list(z[r]["trials"]) == trials1
        assert np.allclose(z[r]["chis"], chis1, rtol=1e-6, atol=0) and np.allclose(z[r]["lams"], lams1, rtol=1e-6, atol=0)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_lm_with_huber_matches_single_rank_and_oracle(tmp_path, world):
    """BASELINE.json config 5 in its stated form at test size: Huber (delta 1, 5 % outliers) Levenberg-Marquardt with
    per-iteration lambda damping on N ranks -- sharded buildSystem / Schur / subtree-distributed Cholesky with the collectives
    inside the library (g2ohip_solve_sharded, *_sharded scalars), device-resident update / push / pop per shard.  The lambda
    sequence, the trials per iteration and chi2 must equal the single-rank run and the oracle-driven loop
    (optimization_algorithm_levenberg.cpp:57-172, base_binary_edge.hpp:92-112)."""
    import torch.multiprocessing as mp
    from openslam_g2o_amd import lm
    from tests.test_gpu_lm import OracleBAGraph, OracleSolverAdapter
    P, L, huber, outliers, n_it = 400, 3600, 1.0, 0.05, 6
    mp.spawn(_lm_worker, args=(world, _free_port(), P, L, huber, outliers, n_it, str(tmp_path)), nprocs=world, join=True)
    pr = ba_case(P, L, outlier_frac=outliers)
    s1, g1 = lm.setup_device_ba(pr, huber_delta=huber)
    done1, chis1, lams1, trials1 = lm.optimize(g1, s1, n_it, "lm")
    og = OracleBAGraph(pr, huber)
    done_o, chis_o, lams_o, trials_o = lm.optimize(og, OracleSolverAdapter(og.o), n_it, "lm")
    assert done1 == done_o and trials1 == trials_o and np.allclose(chis1, chis_o, rtol=1e-6, atol=0)
    cams1, pts1 = s1.baGetEstimates()
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "lm%d.npz" % r))
        assert int(z["done"]) == done1 and list(z["trials"]) == trials1       # same accept / reject decisions on every rank
        assert int(z["collectives"]) == 2        # (a band: the boundary blocks ride with the subtree roots, sharded_merge)
        assert np.allclose(z["chis"], chis1, rtol=1e-6, atol=0) and np.allclose(z["chis"], chis_o, rtol=1e-6, atol=0)
        assert np.allclose(z["lams"], lams1, rtol=1e-6, atol=0)
        # estimates: the landmarks a rank owns, the cameras it owns / shares / needs (halo)
        assert relerr(z["pts"], pts1[z["lm_index"]]) < 1e-6
        valid = (z["pose_owner"] == r) | (z["pose_owner"] < 0)
        valid[z["halo"]] = True
        free = pr["cam_hidx"] >= 0
        cam_ok = np.ones(len(pr["cam_hidx"]), bool)
        cam_ok[free] = valid[pr["cam_hidx"][free]]
        assert relerr(z["cams"][cam_ok], cams1[cam_ok]) < 1e-6
    assert chis1[-1] < chis1[0]


@pytest.mark.parametrize("world", [2, 4])
def test_peer_mailbox_exchange_equals_the_staged_one(tmp_path, world):
    """g2ohip_comm_init_peer (opt-in): the all-reduces of the sharded solve as stores into the peers' device mailboxes (hipIpc handles)
    plus a wait-and-add kernel, instead of RCCL or the host staging.  The ranks are processes sharing one GPU here (what can be
    checked on this box: handles, sequence numbers, parities, slot sums); the Huber LM trajectory, the trial counts and the estimates
    equal the ones of the staged exchange."""
    import torch.multiprocessing as mp
    P, L, huber, outliers, n_it = 400, 3600, 1.0, 0.05, 5
    z = {}
    for kind in ("host", "peer"):
        d = os.path.join(str(tmp_path), kind)
        os.makedirs(d)
        # (graph replay on: with the mailboxes the whole sharded solve, the two launches of every all-reduce included, is ONE captured
        # graph -- the sequence numbers live in device memory)
        mp.spawn(_lm_worker, args=(world, _free_port(), P, L, huber, outliers, n_it, d, -1, "use_graph=1", kind), nprocs=world, join=True)
        z[kind] = [np.load(os.path.join(d, "lm%d.npz" % r)) for r in range(world)]
    for r in range(world):
        a, b = z["host"][r], z["peer"][r]
        assert int(a["done"]) == int(b["done"]) and list(a["trials"]) == list(b["trials"])
        assert np.allclose(a["chis"], b["chis"], rtol=1e-9, atol=0) and np.allclose(a["lams"], b["lams"], rtol=1e-9, atol=0)
        assert relerr(b["pts"], a["pts"]) < 1e-8 and relerr(b["cams"], a["cams"]) < 1e-8
        assert int(b["collectives"]) == 2
    # every rank formed the same sums (slots added in rank order): the shared quantities are bit-identical across the ranks
    for r in range(1, world):
        assert np.array_equal(z["peer"][r]["chis"], z["peer"][0]["chis"]) and np.array_equal(z["peer"][r]["lams"], z["peer"][0]["lams"])


@pytest.mark.parametrize("world", [2, 4])
def test_two_collectives_per_sharded_solve_equal_three(tmp_path, world):
    """Option sharded_merge (default 1): the all-reduce of the boundary blocks / b_p is folded into the one of the subtree roots when
    only the shared top of the tree consumes them (the Schur off-diagonal contributions are what crosses ranks) -- two
    latency-sized collectives per solve instead of three.  Same LM trajectory and estimates as with three (sharded_merge = 0)."""
    import torch.multiprocessing as mp
    P, L, n_it = 640, 5000, 4
    runs = {}
    for merge in (1, 0):
        d = tmp_path / ("m%d" % merge)
        d.mkdir()
        mp.spawn(_lm_worker, args=(world, _free_port(), P, L, 1.0, 0.05, n_it, str(d), -1, "sharded_merge=%d" % merge), nprocs=world, join=True)
        runs[merge] = [np.load(os.path.join(str(d), "lm%d.npz" % r)) for r in range(world)]
    for r in range(world):
        a, b = runs[1][r], runs[0][r]
        assert int(a["collectives"]) == 2 and int(b["collectives"]) == 3
        assert int(a["done"]) == int(b["done"]) and list(a["trials"]) == list(b["trials"])
        assert np.allclose(a["chis"], b["chis"], rtol=1e-9, atol=0) and np.allclose(a["lams"], b["lams"], rtol=1e-9, atol=0)
        assert relerr(a["pts"], b["pts"]) < 1e-9
    assert runs[1][0]["chis"][-1] < runs[1][0]["chis"][0]


def test_start_up_self_test_falls_back_to_the_reference_schedule_on_every_rank(tmp_path):
    """The first sharded solve of a structure runs twice -- as configured (two merged collectives) and on the three-collective,
    uncaptured reference schedule -- and the job continues on the reference schedule if the two disagree (option sharded_selftest,
    block_solver.hip: solve_sharded): here rank 0 is told to corrupt its copy of the first solution (sharded_selftest_break), the
    verdict is a maximum over the ranks, so BOTH ranks fall back: three collectives per solve from then on, and the LM run equals
    the undisturbed one (two collectives: the self-test passed and the configured schedule stayed)."""
    import torch.multiprocessing as mp
    world, P, L, n_it = 2, 640, 5000, 4
    runs = {}
    for tag, opts in (("ok", "sharded_merge=1"), ("broken", "sharded_merge=1,sharded_selftest_break=1"), ("off", "sharded_merge=1,sharded_selftest=0,sharded_selftest_break=1")):
        d = tmp_path / tag
        d.mkdir()
        mp.spawn(_lm_worker, args=(world, _free_port(), P, L, 1.0, 0.05, n_it, str(d), -1, opts), nprocs=world, join=True)
        runs[tag] = [np.load(os.path.join(str(d), "lm%d.npz" % r)) for r in range(world)]
    for r in range(world):
        a, b, c = runs["ok"][r], runs["broken"][r], runs["off"][r]
        assert int(a["collectives"]) == 2 and int(b["collectives"]) == 3 and int(c["collectives"]) == 2    # (off: nothing is compared)
        assert int(a["done"]) == int(b["done"]) and list(a["trials"]) == list(b["trials"])
        assert np.allclose(a["chis"], b["chis"], rtol=1e-9, atol=0) and np.allclose(a["lams"], b["lams"], rtol=1e-9, atol=0)
        assert relerr(a["pts"], b["pts"]) < 1e-9 and relerr(a["pts"], c["pts"]) < 1e-9


def _lm_classes_worker(rank, world, port, P, L, n_it, out_dir):
    import torch
    import torch.distributed as dist
    from openslam_g2o_amd import distributed as D, lm
    from tests.test_gpu_edge_classes import CLASSES, _class_problem
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        pr = _class_problem(P, L, 5)
        pr["classes"] = CLASSES
        s = D.ShardedBlockSolver(6, 3, rank=rank, world=world, comm=D.HostStagedComm(world), mode="subtree")
        s.setup_ba(pr, torch_device=torch.device("cuda", 0), fused=True)
        assert s.attach_library_comm("host")
        done, chis, lams, trials = lm.optimize(D.ShardedBAGraph(s), s, n_it, "lm")
        cams, pts = s.local.baGetEstimates()
        np.savez(os.path.join(out_dir, "c%d.npz" % rank), done=done, chis=chis, lams=lams, trials=trials, pts=pts, lm_index=s.lm_index)
    finally:
        dist.destroy_process_group()


def test_sharded_lm_with_edge_classes_matches_the_single_rank_run(tmp_path):
    """Two CameraParameters and three robust kernels in one edge set (edge classes), sharded over two ranks: every rank holds the class
    table, its observations carry their class; the LM trajectory equals the single-rank device run (itself held to the oracle by
    tests/test_gpu_edge_classes.py)."""
    import torch.multiprocessing as mp
    from openslam_g2o_amd import capi, lm
    from tests.test_gpu_edge_classes import _class_problem, _device
    world, P, L, n_it = 2, 400, 3600, 5
    mp.spawn(_lm_classes_worker, args=(world, _free_port(), P, L, n_it, str(tmp_path)), nprocs=world, join=True)
    pr = _class_problem(P, L, 5)
    s1, g1, _ = _device(pr)
    done1, chis1, lams1, trials1 = lm.optimize(g1, s1, n_it, "lm")
    _, pts1 = s1.baGetEstimates()
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "c%d.npz" % r))
        assert int(z["done"]) == done1 and list(z["trials"]) == trials1
        assert np.allclose(z["chis"], chis1, rtol=1e-6, atol=0) and np.allclose(z["lams"], lams1, rtol=1e-6, atol=0)
        assert relerr(z["pts"], pts1[z["lm_index"]]) < 1e-6
    assert chis1[-1] < chis1[0]


def test_library_comm_over_rccl_single_rank():
    """g2ohip_comm_init_rccl / ncclAllReduce inside the library on the one GPU a test box has (world 1): the sharded solve
    through RCCL equals the plain solve."""
    import torch
    from openslam_g2o_amd import distributed as D
    pr = ba_case(300, 3000)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    s = D.ShardedBlockSolver(6, 3, rank=0, world=1, mode="subtree", force_exchange=True)
    s.setup_ba(pr, torch_device=dev, fused=True)
    assert s.attach_library_comm("rccl")
    s.buildSystem()
    chi = s.chi2()
    md = s.maxDiagonal()
    s.setLambda(7.0, True)
    assert s.solve()
    sc = s.computeScale(7.0)
    s.restoreDiagonal()
    x = s.local.x()
    o = oracle_ba(pr)
    o.build_system()
    assert abs(chi - o.chi2()) <= 1e-9 * chi and abs(md - o.max_diagonal()) <= 1e-12 * md
    o.set_lambda(7.0, True)
    assert o.solve()
    assert relerr(x, o.x()) < dx_tolerance(o)[0] and abs(sc - o.compute_scale(7.0)) <= 1e-6 * abs(sc)


@pytest.mark.parametrize("ranks,poses,comm", [(2, 3000, "staged"), (8, 8000, "staged"), (4, 4000, "peer")])
def test_bench_launches_its_own_ranks(ranks, poses, comm):
    """Plain `python bench.py --gpus N` (no launcher in front, WORLD_SIZE unset) starts one rank per GPU itself; on this
    1-GPU box the ranks share cuda:0 and the exchange is staged through gloo.  Rank 0 prints the one JSON line with
    the collectives' kind, a kernel table per rank and the three all-reduce times.  N = 8 is the shape of the driver's scaling
    run: rank -> device mapping, the partition at world 8 (every rank owns poses) and the JSON line are proven here, and the
    gathered pose increment is held to the CPU oracle on the whole graph (--check-oracle).  `--comm peer`: the same through the
    library's peer-mailbox exchange (the mailboxes of all ranks on the one device here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--comm", comm, "--poses", str(poses),
                        "--landmarks", str(10 * poses), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--check-oracle"], env=env,
                       cwd=root, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [q for q in r.stdout.splitlines() if q.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == ranks and out["solve_ok"] and out["collectives"] == ("peer" if comm == "peer" else "host")
    assert len(out["per_rank_kernel_ms"]) == ranks
    assert len(out["all_reduce_ms"]) == 3 and all(len(v) == ranks and v[0] is not None for v in out["all_reduce_ms"].values())
    per_rank = out["shard"]["poses_per_rank"]          # [shared, rank 0, rank 1, ...]
    assert len(per_rank) == ranks + 1 and all(n > 0 for n in per_rank[1:]) and sum(per_rank) == out["config"]["poses"] - 2
    assert out["dx_pose_rel_err"] < 1e-8
