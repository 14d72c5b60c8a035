#!/usr/bin/env python
"""bench.py -- linear-solve iteration benchmark of the MI355X-native g2o block solver.

One "step" = one pass of the hot path over the synthetic BA graph:
  buildSystem (K1-K3)  ->  setLambda(backup) (K4)  ->  solve (K5-K13: Schur complement,
  multifrontal block Cholesky of the reduced pose system, landmark back-substitution)
  ->  restoreDiagonal
i.e. the linear algebra of one Levenberg-Marquardt trial
(/root/reference/g2o/core/optimization_algorithm_levenberg.cpp:95-142), with the per-edge
Jacobians / information / errors already resident in HBM when the timed region starts.

Workload at every N: BASELINE.json's metric configuration -- 100 000 poses / 1 000 000
landmarks / 5 000 000 observations (configs[3]; it fits one GPU).  For N > 1 the landmarks
(and their edges) are sharded across ranks and the Schur contributions are summed with an
RCCL all-reduce (strong scaling, see openslam_g2o_amd/distributed.py and DESIGN.md).

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy peak)
MFMA_F64_PEAK_TFLOPS = 78.6   # dense fp64 matrix rate: 256 CUs x 4 SIMDs x (v_mfma_f64_16x16x4_f64 = 2048 flops / 64 cycles) x 2.4 GHz


def algorithmic_bytes(E, P, L, S, pp_nnzb, nnzL, p=6, l=3, d=2, folded=False):
    """Compulsory traffic per kernel / stage (8 B per double, 4 B per index), SURVEY.md 8d."""
    kb = {}
    kb["assemble_vertex(pose)"] = E * 8 * (d * p + d * d + d) + P * 8 * (p * p + p)
    kb["assemble_vertex(landmark)"] = E * 8 * (d * l + d * d + d) + L * 8 * (l * l + l)
    kb["assemble_offdiag(Hpl)"] = E * 8 * (d * p + d * l + d * d) + E * 8 * p * l
    # fused EdgeProjectXYZ2UV assembly (no Jacobian arrays): measurements + information in, blocks out
    # (information = identity is declared for the whole set, so it is not read per edge)
    kb["fused:assemble_vertex(landmark)"] = E * 8 * (d + p * l + d) + L * 8 * (l + l * l + l) + P * 8 * 12
    kb["fused:assemble_vertex(pose)"] = E * 8 * (d + l) + P * 8 * (12 + p * p + p)
    kb["landmark_inverse"] = L * 8 * (2 * l * l + 2 * l)
    # Hpl, Hll, b_l in; Dinv (the landmark inversion is done on the staged tile) and one Hschur worth of partials out
    kb["schur_tiles"] = E * 8 * p * l + L * 8 * (l * l + l) + L * 8 * l * l + S * 8 * p * p
    kb["schur_reduce"] = pp_nnzb * 8 * p * p + 2 * S * 8 * p * p + 3 * P * 8 * p      # Hpp + partials in, Hschur + bschur out
    kb["back_substitute"] = E * 8 * p * l + L * 8 * (l * l + 2 * l) + P * 8 * p
    n = p * P
    nnz_up = S * p * p  # upper blocks of Hschur (diagonal blocks counted full)
    # the forward sweep is fused into the factorisation (it reads b and writes y, L never leaves LDS in between);
    # the backward sweep reads L once
    kb["chol_factor(all levels)"] = 8 * nnz_up + 8 * nnzL + 16 * n
    kb["chol_solve(all levels)"] = 8 * nnzL + 24 * n
    if folded:
        # one GPU: the reduction pass of the Schur complement is folded into the factorisation's front assembly --
        # the factor kernel reads Hpp and (at least one Hschur worth of) partial blocks, Hschur is never written;
        # the "schur_reduce" slot only forms bschur
        kb["schur_reduce"] = 3 * P * 8 * p
        kb["chol_factor(all levels)"] += pp_nnzb * 8 * p * p
    stage = {
        "B_asm": E * (8 * (d * p + d * l + d * d + d) + 8) + E * 8 * p * l + P * 8 * (p * p + p) + L * 8 * (l * l + l),
        "B_schur": (E * 8 * p * l + L * 8 * (l * l + l) + P * 8 * (p * p + p)) + (L * 8 * l * l + S * 8 * p * p + P * 8 * p),
        "B_chol": 8 * nnz_up + 8 * nnzL + 16 * nnzL + 24 * n,
        "B_back": E * 8 * p * l + L * 8 * (l * l + 2 * l) + P * 8 * p + L * 8 * l,
    }
    if folded:
        stage["B_schur"] = E * 8 * p * l + L * 8 * (l * l + l) + L * 8 * l * l + S * 8 * p * p + 3 * P * 8 * p
        stage["B_chol"] += pp_nnzb * 8 * p * p
    return kb, stage


def cpu_baseline(prob, lam, want_x=True, include_linearize=True, omp=False, reps=3):
    """The oracle (CPU restatement of the reference path) timed on this host for `reps` full iterations of the
    same workload; medians per stage.  omp=False: one thread = the reference's default build (G2O_USE_OPENMP OFF,
    CMakeLists.txt:137); omp=True: the reference's optional OpenMP regions (buildSystem over edges,
    block_solver.hpp:527; Schur complement over landmarks, :379) on every host core -- the sparse Cholesky stays
    serial there too (CSparse)."""
    from oracle import oracle as O
    o = O.OracleSolver(6, 3, prob["nP"], prob["nL"], True, omp=omp)
    k = o.add_edge_set(2, prob["v0"], prob["v1"])
    o.set_dims(k, 3, 6)
    t0 = time.perf_counter()
    o.build_structure()
    t_struct = time.perf_counter() - t0
    t_asm, t_solve, t_schur, t_lin, t_num = [], [], [], [], []
    ok = True
    for rep in range(reps + 1):            # rep 0 carries the one-time ordering + symbolic step: not counted
        tl = 0.0
        if include_linearize:              # the reference's buildSystem runs linearizeOplus per edge (block_solver.hpp:531)
            t0 = time.perf_counter()
            Jp, Jc, err = O.ba_edges(prob["cams"], prob["pts"], prob["cam_idx"], prob["pt_idx"], prob["meas"], prob["f"],
                                     prob["cx"], prob["cy"], omp=omp)
            tl = time.perf_counter() - t0
            o.set_edge_data(k, Jp, Jc, prob["omega"], err)
        elif rep == 0:
            o.set_edge_data(k, prob["Jp"], prob["Jc"], prob["omega"], prob["err"])
        t0 = time.perf_counter()
        o.build_system()
        ta = time.perf_counter() - t0 + tl
        o.set_lambda(lam, True)
        t0 = time.perf_counter()
        ok = o.solve() and ok              # Schur + numeric Cholesky + back-substitution
        ts = time.perf_counter() - t0
        o.restore_diagonal()
        if rep > 0:
            tt = o.times()
            t_asm.append(ta); t_solve.append(ts); t_schur.append(tt["schur"]); t_lin.append(tt["linear"]); t_num.append(tt["numeric"])
    med = lambda v: float(np.median(v))
    out = dict(ok=bool(ok), t_structure=t_struct, t_assembly=med(t_asm), t_solve=med(t_solve), t_schur=med(t_schur),
               t_linear=med(t_lin), t_numeric=med(t_num), lnz=o.lnz(), reps=reps, threads=o.L.orc_num_threads(),
               spread=float((max(a + b for a, b in zip(t_asm, t_solve)) - min(a + b for a, b in zip(t_asm, t_solve))) /
                            np.median([a + b for a, b in zip(t_asm, t_solve)])))
    o.set_lambda(lam, True)
    x = o.x() if want_x else None
    o.restore_diagonal()
    return out, x


def host_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--poses", type=int, default=100000)
    ap.add_argument("--landmarks", type=int, default=1000000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["chain", "grid"], default="chain",
                    help="chain: BASELINE.json's metric configuration (a camera trajectory: the reduced system is a band); grid: cameras on a "
                         "square lattice, every point seen by all cameras within a radius (synthetic.make_ba_grid: a two-dimensional mesh "
                         "with real fill -- a parity / sizing workload next to the headline, --poses 10000 / 50000; --landmarks = 10 per "
                         "camera unless given)")
    ap.add_argument("--nd-leaf", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="solver option name=value (tuning experiments)")
    ap.add_argument("--edge-data", choices=["fused", "arrays"], default="fused",
                    help="fused: estimates + measurements resident in HBM, errors/Jacobians evaluated inside buildSystem "
                         "(what the reference's buildSystem does per edge); arrays: precomputed Jacobian arrays resident in HBM")
    ap.add_argument("--information", choices=["identity", "edge"], default="identity",
                    help="identity: information().setIdentity() declared for the whole edge set, as SURVEY.md 8d's workload has it (the fused "
                         "kernels then do not read it); edge: a different symmetric positive definite 2 x 2 information matrix per edge, "
                         "read per observation by every kernel (and by the CPU baseline)")
    ap.add_argument("--comm", choices=["rccl", "staged", "peer"], default="rccl",
                    help="staged: gloo + host staging with every rank on cuda:0 (functional check of the N>1 path on a "
                         "1-GPU box; its timing is meaningless); peer: the library's opt-in peer-mailbox exchange "
                         "(g2ohip_comm_init_peer: hipIpc handles, stores over xGMI) instead of RCCL for the solve's all-reduces")
    ap.add_argument("--check-oracle", action="store_true", help="N > 1: gather the pose increment of the last solve on rank 0 and compare it "
                    "with one CPU oracle iteration on the whole graph (dx_pose_rel_err in the JSON line); for sizes the oracle "
                    "finishes in seconds")
    ap.add_argument("--emulate", default="", help="R/W: run rank R of a W-rank job alone with a no-op exchange (results are "
                    "meaningless, per-rank kernel and wall time without communication are not) -- sizing tool for 1-GPU boxes")
    ap.add_argument("--graph", choices=["on", "off"], default="on",
                    help="hipGraph replay of the two launch-bound kernel sequences (factorisation + fused forward sweep, "
                         "backward sweep: one launch per tree level); the HIP timing events sit between the graphs")
    ap.add_argument("--mode", choices=["auto", "subtree", "replicated", "pcg"], default="auto", help="N>1 reduced-solve strategy")
    ap.add_argument("--dump-xp", default="", help="rank 0 saves the pose increment to this .npy (cross-run comparison)")
    args = ap.parse_args()

    from openslam_g2o_amd.launch import relaunch_if_needed
    relaunch_if_needed(args.gpus, __file__)       # plain `python bench.py --gpus N`: one rank per GPU via torch.distributed.run

    import torch
    from openslam_g2o_amd import capi, synthetic as S
    from openslam_g2o_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    emulate = None
    if args.emulate:
        emulate = tuple(int(v) for v in args.emulate.split("/"))
    comm_note = None
    comm_asked = args.comm
    if args.comm == "peer":      # process group and rank -> device mapping as for rccl; only the library's communicator differs
        args.comm = "rccl"
    if world > 1 and args.comm == "rccl" and torch.cuda.device_count() < world:
        # fewer GPUs than ranks (a 1-GPU box): RCCL refuses two ranks per device -> functional run through host staging
        comm_note = "%s requested, %d GPU(s) visible for %d ranks: ranks share cuda:0, %s" % (
            comm_asked, torch.cuda.device_count(), world,
            "mailboxes on the one device" if comm_asked == "peer" else "exchange staged through gloo")
        if rank == 0:
            sys.stderr.write("bench: " + comm_note + "\n")
        args.comm = "staged"
    if args.comm == "staged":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.comm == "staged":
            dist.init_process_group("gloo")
            comm = D.HostStagedComm(world)
        else:
            dist.init_process_group("nccl", device_id=dev)

    P, L = args.poses, args.landmarks
    if args.workload == "grid":
        ppc = max(L // P, 1) if L != 1000000 or P == 100000 else 10
        prob = S.make_ba_grid(P, pts_per_cam=ppc)
        P, L = prob["P"], prob["L"]
    else:
        prob = S.make_ba_problem(P, L)                  # identical on every rank (counter-based RNG)
    fused = args.edge_data == "fused"
    prob["omega"] = S.ba_omega(prob)
    if args.information == "edge":
        rng = np.random.RandomState(7)
        a, b_, c = rng.uniform(0.5, 2.0, prob["E"]), rng.uniform(0.5, 2.0, prob["E"]), rng.uniform(-0.3, 0.3, prob["E"])
        prob["omega"] = np.ascontiguousarray(np.stack([a, c, c, b_], axis=1))   # column-major 2 x 2, eigenvalues >= 0.2
        prob["info"] = prob["omega"]
    if not fused:
        Jp, Jc, err = S.ba_linearize(prob)
        prob.update(Jp=Jp, Jc=Jc, err=err)
    lam = 1e-5 * 1.0e6                                  # tau * max diag(H) order of magnitude; fixed for reproducibility

    if emulate:
        class NullComm(D.TorchComm):
            def all_reduce_sum(self, tensors):
                pass

            def all_reduce_scalar(self, v, device="cpu"):
                return v
        solver = D.ShardedBlockSolver(6, 3, rank=emulate[0], world=emulate[1], device=local_rank, comm=NullComm(1),
                                      mode="subtree" if args.mode == "auto" else args.mode)
    else:
        solver = D.ShardedBlockSolver(6, 3, rank=rank, world=world, device=local_rank, comm=comm, mode=args.mode)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        solver.local.setOption(k_, float(v_))
    use_graph = args.graph == "on"
    side = torch.cuda.Stream(device=dev)          # graphs cannot be captured on the default stream
    torch.cuda.set_stream(side)
    shard = solver.setup_ba(prob, torch_device=dev, nd_leaf=args.nd_leaf, fused=fused)
    if use_graph:
        solver.local.setOption("use_graph", 1)
    solver.local.setProfiling(True)
    # N > 1: the collectives run INSIDE libg2ohip (ncclCommInitRank / ncclAllReduce bound there; g2ohip_solve_sharded);
    # falls back to the torch.distributed formulation of the same exchange if the library communicator cannot be
    # brought up (the reason goes to stderr and into the JSON line)
    lib_comm = "n/a"
    world_checked = None     # what an all-reduce of ones through the library's communicator returned (= the ranks that took part)
    if emulate and solver.mode == "subtree" and solver._native_exchange():
        # the library's own sharded solve (g2ohip_solve_sharded: every phase queued by one call) with the all-reduces skipped
        solver.local.setOption("comm_emulate", 1)
        solver._lib_comm = lib_comm = "none (emulation: all-reduces skipped)"
    if (world > 1 or emulate) and solver.mode == "subtree" and not emulate:
        try:
            lib_comm = "peer" if comm_asked == "peer" else ("rccl" if args.comm == "rccl" else "host")
            if not solver.attach_library_comm(lib_comm):
                lib_comm = "torch (library communicator not applicable)"
            else:
                # self-check: an all-reduce of ones must give the world size on every rank
                one = torch.ones(8, dtype=torch.float64, device=dev)
                solver.local.commAllReduce(one.data_ptr(), 8, 0)
                torch.cuda.synchronize()
                if not bool((one == float(world)).all().item()):
                    raise RuntimeError("library all-reduce self-check failed: %r" % one.tolist())
                world_checked = int(one[0].item())
        except Exception as e:      # noqa: BLE001
            sys.stderr.write("bench: library communicator unavailable (%s); using torch.distributed\n" % e)
            solver._lib_comm = None
            lib_comm = "torch (fallback: %s)" % type(e).__name__

    def step():
        solver.buildSystem()
        solver.setLambda(lam, True)
        ok = solver.solve()
        solver.restoreDiagonal()
        return ok

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up with every kernel slot timed: finds the dominant slot; the timed region then carries HIP events
    # around that slot only (event records are not free: ~20 per iteration cost ~0.1 ms of a 2.6 ms iteration)
    ok = True
    n_find = max(args.warmup - 1, 2)                       # warm-up steps with every slot timed: the LAST one finds the dominant kernel ...
    for i in range(n_find):
        if i == n_find - 1:
            solver.local.kernelTimes(reset=True)           # (the first steps carry one-time costs: code loading, graph capture)
        ok = step() and ok
    wt = solver.local.kernelTimes(reset=True)
    slot_names = [solver.local.L.g2ohip_kernel_name(k).decode() for k in range(solver.local.L.g2ohip_kernel_slots())]
    # dominant KERNEL: the slot with the largest time per launch (a slot is one kernel; the triangular sweeps are several)
    dom_name = max(wt.items(), key=lambda kv: kv[1][0] / max(kv[1][1], 1))[0] if wt else "schur_tiles"
    # N > 1 (and its emulation): no kernel timers inside the timed region -- the library then replays the whole sharded solve
    # as ONE hipGraph where nothing has to cross the host (option sharded_graph); the per-kernel table comes from the second pass
    no_events = (world > 1 or emulate) and solver.mode == "subtree"
    solver.local.setProfiling(False if no_events else 2 + slot_names.index(dom_name))
    for _ in range(max(args.warmup - n_find, 1)):          # ... and the rest in the mode of the timed region (its launch graphs
        ok = step() and ok                                 # are captured here, not inside the timed region)
    solver.local.kernelTimes(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ok = step() and ok
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.comm == "rccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = 1e3 * dt / args.steps
    dom_timed = {} if no_events else solver.local.kernelTimes(reset=True)   # the dominant slot, from inside the timed region
    solver.local.setProfiling(True)                            # second pass, not timed: the whole per-kernel table
    for _ in range(args.steps):
        ok = step() and ok
    barrier()

    ktimes = solver.local.kernelTimes(reset=True)
    ktimes.update(dom_timed)
    rank_tables = None
    if world > 1:
        mine = {name: round(1e3 * tot / n, 5) for name, (tot, n) in ktimes.items()}
        rank_tables = [None] * world
        dist.all_gather_object(rank_tables, mine)
    xp_all = None
    if args.check_oracle and world > 1:
        xp_all = solver.gather_x_poses()            # collective: every rank takes part (x of the last damped solve)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    st = solver.local.stats()
    S_blocks = solver.local.nnzb(capi.HSCHUR)
    pp_nnzb = solver.local.nnzb(capi.HPP)
    E_loc, L_loc = shard["E_local"], shard["L_local"]
    folded = world == 1 and not emulate and not any(kv.replace(" ", "") in ("fuse_schur_reduce=0", "linear_solver=1") for kv in args.opt)
    kb, stage_b = algorithmic_bytes(E_loc, prob["nP"], L_loc, S_blocks, pp_nnzb, st["choleskyNNZ"], folded=folded)
    if st.get("bandChains", 0) > 0:
        # the factorisation is two kernels: the leaf chains of the band (band_wave_kernel) and the tree levels above them
        # (wave_front_kernel).  Its compulsory bytes are split by what each one factorises: nnz(L) exactly, the matrix
        # entries and the forward sweep's vectors by the share of the pivot columns.
        n_sc = 6 * prob["nP"]
        tot = kb["chol_factor(all levels)"]
        band = 8 * st["bandCholeskyNNZ"] + (tot - 8 * st["choleskyNNZ"]) * st["bandPivots"] / n_sc
        kb["chol_factor(band chains)"] = band
        kb["chol_factor(all levels)"] = tot - band
    om_bytes = 0 if args.information == "identity" else 8 * 4      # the per-observation information matrix, where it is read
    fused_kb = {}
    if fused:
        kb["assemble_vertex(landmark)"] = kb["fused:assemble_vertex(landmark)"]
        kb["assemble_vertex(pose)"] = kb["fused:assemble_vertex(pose)"] + E_loc * om_bytes
        # The fused kernels do not read Hpl: what they move by construction (estimates, measurements, results; index tables
        # excluded like everywhere) is far below the un-fused stage formulas of SURVEY.md 8d, which stay the yardstick of
        # `roofline` (work done per second).  back_substitute is reported on ITS OWN bytes -- on the un-fused formula
        # (an Hpl read it does not make) the slot printed more than the HBM peak.
        p_, l_, d_ = 6, 3, 2
        fused_kb["schur_tiles"] = E_loc * (8 * d_ + om_bytes) + L_loc * 8 * (l_ + l_ * l_ + l_) + prob["nP"] * 8 * 12 + S_blocks * 8 * p_ * p_
        fused_kb["back_substitute"] = E_loc * (8 * d_ + om_bytes) + L_loc * 8 * (l_ + l_ * l_ + 2 * l_) + prob["nP"] * 8 * (12 + p_)
        kb["back_substitute"] = fused_kb["back_substitute"]
    per_kernel = {}
    for name, (tot, n) in ktimes.items():
        avg = tot / n
        per_kernel[name] = dict(avg_ms=1e3 * avg, launches_per_step=n / args.steps,
                                algorithmic_GB=kb.get(name, 0) / 1e9,
                                achieved_GBs=(kb.get(name, 0) / 1e9 / avg) if avg > 0 else 0.0)
        if name in fused_kb:
            per_kernel[name]["fused_algorithmic_GB"] = fused_kb[name] / 1e9   # bytes this (fused) kernel moves by construction
    dname = dom_name if dom_name in per_kernel else max(per_kernel.items(), key=lambda kv: kv[1]["avg_ms"] * kv[1]["launches_per_step"])[0]
    dk = per_kernel[dname]
    # HBM bytes per launch from the PMC counters: collected by a separate rocprofv3 --pmc pass of this
    # same command (profiles/r1_pmc_traffic.json, recipe in its _note); null when that file is absent
    # or the workload differs from the profiled one.
    traffic = None
    pmc_all = {}
    pmc_path = next((q for q in (os.path.join(ROOT, "profiles", "r%d_pmc_traffic.json" % r) for r in (9, 8, 7, 6, 5, 4, 3, 2, 1))
                     if os.path.exists(q)), "")
    if pmc_path and world == 1 and not emulate and (P, L) == (100000, 1000000):   # (the counters were taken on the whole graph on one GPU)
        pmc_all = json.load(open(pmc_path))
        pmc = pmc_all.get(dname)
        if pmc:
            traffic = pmc["hbm_bytes_per_step"]
        # next to the rate of WORK (un-fused algorithmic bytes / time, which a fused kernel can push past what any copy
        # reaches) the rate of TRAFFIC the counters saw for the same kernel: bytes actually moved / time
        for name, row in per_kernel.items():
            q = pmc_all.get(name)
            if isinstance(q, dict) and row["avg_ms"] > 0:
                per_launch = q["hbm_bytes_per_step"] / max(row["launches_per_step"], 1e-9)   # (per launch of the SLOT, like avg_ms)
                row["pmc_traffic_GB"] = per_launch / 1e9
                row["pmc_GBs"] = per_launch / 1e9 / (1e-3 * row["avg_ms"])
                # matrix cores: v_mfma_f64_16x16x4_f64 wave instructions counted by SQ_INSTS_MFMA (2048 flops each) over this
                # run's time for the slot, against the dense fp64 matrix peak; the busy-cycle counter beside it
                n_mfma = q.get("SQ_INSTS_MFMA_per_step", 0)
                if n_mfma:
                    step_s = 1e-3 * row["avg_ms"] * row["launches_per_step"]
                    row["mfma_TFLOPs"] = 2048.0 * n_mfma / step_s / 1e12
                    row["mfma_util"] = row["mfma_TFLOPs"] / MFMA_F64_PEAK_TFLOPS
                    if q.get("SQ_BUSY_CYCLES_per_step"):
                        row["mfma_busy_cycles_over_sq_busy_cycles"] = q.get("SQ_VALU_MFMA_BUSY_CYCLES_per_step", 0) / q["SQ_BUSY_CYCLES_per_step"]
    roofline = dict(kernel=dname, bound="hbm", achieved=dk["achieved_GBs"], peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=dk["achieved_GBs"] / HBM_PEAK_GBS, traffic=traffic,
                    # the same fraction on the bytes the counters saw move (traffic / this run's launch time / peak)
                    frac_traffic=(traffic / (1e-3 * dk["avg_ms"]) / 1e9 / HBM_PEAK_GBS) if traffic and dk["avg_ms"] > 0 else None,
                    algorithmic_bytes_per_launch=kb.get(dname, 0), avg_launch_ms=dk["avg_ms"],
                    traffic_source="profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, 2x FETCH correction)" % os.path.basename(pmc_path)
                    if traffic else None,
                    # the counters come from a separate rocprofv3 pass (gpurun refuses --pmc next to other tracing), taken at this commit
                    traffic_commit=(pmc_all.get("_commit") if traffic else None))
    # the factorisation on its FLOPS (dense-front count of the symbolic analysis, g2ohip_stats.choleskyFlops) over the time of its slots:
    # where it is the dominant kernel (graphs with real fill: --workload grid) the matrix cores bound the path, not HBM
    fl = float(st.get("choleskyFlops", 0.0))
    t_factor = sum(1e-3 * v["avg_ms"] * v["launches_per_step"] for k, v in per_kernel.items() if k.startswith("chol_factor"))
    if fl > 0 and t_factor > 0:
        for k, v in per_kernel.items():
            if k.startswith("chol_factor"):
                v["factor_TFLOPs(all factor slots)"] = fl / t_factor / 1e12
        if dname.startswith("chol_factor"):
            roofline = dict(kernel=dname, bound="mfma", achieved=fl / t_factor / 1e12, peak=MFMA_F64_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=fl / t_factor / 1e12 / MFMA_F64_PEAK_TFLOPS, traffic=None, flops_per_launch=fl, avg_launch_ms=1e3 * t_factor,
                            note="dense fp64 matrix peak; flops = dense-front count of this factorisation's symbolic analysis (nnz(L) %d)" % st["choleskyNNZ"],
                            hbm_view=roofline)
    mfma_kernels = {k: {"mfma_util": round(v["mfma_util"], 4), "mfma_TFLOPs": round(v["mfma_TFLOPs"], 2)} for k, v in per_kernel.items() if "mfma_util" in v}
    if mfma_kernels:
        roofline["mfma"] = dict(peak_TFLOPs=MFMA_F64_PEAK_TFLOPS, unit="TFLOP/s (dense fp64 matrix)", kernels=mfma_kernels)

    x_gpu = solver.local.x()
    if args.dump_xp:
        os.makedirs(os.path.dirname(os.path.abspath(args.dump_xp)), exist_ok=True)
        np.save(args.dump_xp, x_gpu[:6 * prob["nP"]])
    out = {
        "metric": "linear-solve ms/iter (buildSystem + setLambda + solve + restoreDiagonal), 100k-pose BA" if args.workload == "chain"
        else "linear-solve ms/iter (buildSystem + setLambda + solve + restoreDiagonal), grid BA (visibility by distance)",
        "value": ms, "unit": "ms/iter", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "iters_per_s": 1e3 / ms, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("synthetic BA (BASELINE.json configs[3]): %d poses / %d landmarks / %d observations, "
                                "BlockSolver_6_3 semantics, Schur + multifrontal block Cholesky, lambda=%g" % (P, L, prob["E"], lam))
                   if args.workload == "chain" else
                   ("synthetic BA, visibility by distance (NOT the metric configuration): %d cameras on a square lattice / %d points / %d "
                    "observations, BlockSolver_6_3 semantics, Schur + multifrontal block Cholesky, lambda=%g" % (P, L, prob["E"], lam)),
                   "poses": P, "landmarks": L, "edges": prob["E"], "parallelism": solver.parallelism(),
                   "edge_data": "estimates+measurements in HBM, errors/Jacobians evaluated inside buildSystem" if fused
                   else "precomputed Jacobian arrays in HBM",
                   "information": "identity, declared for the whole edge set (not read per edge)" if args.information == "identity"
                   else "one 2 x 2 matrix per edge, read per observation"},
        "solve_ok": bool(ok),
        "launch": ("hipGraph replay of the launch sequences (elimination-tree levels grouped into dependency-driven launches)" if use_graph else "plain launches") +
                  "; timed region: HIP events around the dominant kernel slot only (roofline); the other per-kernel times come "
                  "from a second pass of the same steps with every slot timed",
        "roofline": roofline,
        "kernels": per_kernel,
        "stage_algorithmic_GB": {k: v / 1e9 for k, v in stage_b.items()},
        "solver_stats": {k: st[k] for k in ("choleskyNNZ", "numFronts", "numLevels", "maxFrontDim", "timeSymbolicDecomposition", "bandChains",
                                            "bandCholeskyNNZ", "bandPivots", "choleskyFlops")},
    }
    if emulate:
        out["emulate"] = "rank %d of %d alone, exchange skipped: timing only" % emulate
    if world > 1 or emulate:
        out["collectives"] = lib_comm
        if world_checked is not None:
            out["rccl_world_checked" if lib_comm == "rccl" else "comm_world_checked"] = world_checked
        out["shard"] = dict(rank0_edges=E_loc, rank0_landmarks=L_loc, exchange_doubles_per_solve=solver.exchange_volume(),
                            comm=comm_asked if comm_asked == "peer" else args.comm)
        if comm_note:
            out["shard"]["comm_note"] = comm_note
        if rank_tables:
            out["per_rank_kernel_ms"] = rank_tables      # one table per rank, same slots as "kernels"
            out["all_reduce_ms"] = {k: [t.get(k) for t in rank_tables] for k in sorted(rank_tables[0]) if k.startswith("exchange(")}
        if solver.mode == "subtree":
            out["shard"].update(boundary_blocks=int(len(solver.boundary)), reduced_blocks=int(solver.nnzb_reduced),
                                poses_per_rank=np.bincount(solver.pose_owner + 1).tolist())
    # size-independent correctness property at full size: residual of the damped system
    solver.setLambda(lam, True)
    if world == 1 and not emulate:
        r = solver.local.multiplyHessian(x_gpu) - solver.local.b()
        out["residual_rel"] = float(np.abs(r).max() / np.abs(solver.local.b()).max())
    solver.restoreDiagonal()

    if world == 1 and not emulate and not args.no_cpu_baseline:
        grid = args.workload == "grid"   # (a CPU iteration of the grid graph is minutes, not seconds: one timed repetition, no thread sweep)
        cb, x_cpu = cpu_baseline(prob, lam, include_linearize=fused, reps=1 if grid else 3)
        cpu_ms = 1e3 * (cb["t_assembly"] + cb["t_solve"])
        # "best CPU": the reference's optional OpenMP regions.  Its Schur loop takes a mutex per pose row
        # (block_solver.hpp:411) and neighbouring landmarks hit the same rows, so more threads are not always faster:
        # a few thread counts are tried (one iteration each), the best one is then measured like the 1-thread case
        from oracle import oracle as O_
        ncores = os.cpu_count() or 1
        cands = [] if grid else sorted({t for t in (4, 8, 16, 32, 64, ncores // 2, ncores) if 1 < t <= ncores})
        tried = {}
        for t in cands:
            O_.lib(True).orc_set_num_threads(t)
            c1, _ = cpu_baseline(prob, lam, want_x=False, include_linearize=fused, omp=True, reps=1)
            tried[t] = 1e3 * (c1["t_assembly"] + c1["t_solve"])
        best_t = min(tried, key=tried.get) if tried else 1
        if grid:
            cbo, omp_ms = cb, cpu_ms
        else:
            O_.lib(True).orc_set_num_threads(best_t)
            cbo, _ = cpu_baseline(prob, lam, want_x=False, include_linearize=fused, omp=True)
            omp_ms = 1e3 * (cbo["t_assembly"] + cbo["t_solve"])
        out["cpu_baseline"] = {"value": cpu_ms, "unit": "ms/iter", "cores": 1, "kind": "port",
                               "sample": "median of %d full iterations of the same %d-pose workload, one thread = the reference's "
                                         "default build (assembly %.0f ms + solve %.0f ms, spread %.1f %%; one-time structure %.1f s "
                                         "and ordering/symbolic excluded)" % (
                                             cb["reps"], P, 1e3 * cb["t_assembly"], 1e3 * cb["t_solve"], 100 * cb["spread"], cb["t_structure"]),
                               "breakdown_ms": {"assembly": 1e3 * cb["t_assembly"], "schur": 1e3 * cb["t_schur"],
                                                "linear_solver": 1e3 * cb["t_linear"], "numeric_cholesky": 1e3 * cb["t_numeric"]},
                               "choleskyNNZ": cb["lnz"], "host": host_model(), "host_cores": os.cpu_count(),
                               "best_cpu": {"value": omp_ms, "unit": "ms/iter", "cores": cbo["threads"], "kind": "port",
                                            "sample": "median of %d iterations with the reference's optional OpenMP regions on "
                                                      "(buildSystem over edges, Schur complement over landmarks; the sparse Cholesky "
                                                      "is serial in the reference too)" % cbo["reps"],
                                            "breakdown_ms": {"assembly": 1e3 * cbo["t_assembly"], "schur": 1e3 * cbo["t_schur"],
                                                             "linear_solver": 1e3 * cbo["t_linear"]},
                                            "threads_tried_ms": {str(k): round(v, 1) for k, v in tried.items()}}}
        out["speedup_vs_cpu"] = cpu_ms / ms
        nx = np.abs(x_cpu).max()
        out["dx_rel_err"] = float(np.abs(x_gpu - x_cpu).max() / nx)
        # chi2 after applying each update (host-side error evaluation of the updated state)
        e_g = S.ba_linearize(S.ba_oplus(prob, x_gpu), jac=False)
        e_c = S.ba_linearize(S.ba_oplus(prob, x_cpu), jac=False)
        Om = prob["omega"]

        def chi_of(e):   # sum of e' Omega e (column-major 2 x 2 per edge)
            return float(np.sum(e[:, 0] * (Om[:, 0] * e[:, 0] + Om[:, 2] * e[:, 1]) + e[:, 1] * (Om[:, 1] * e[:, 0] + Om[:, 3] * e[:, 1])))
        chi_g, chi_c = chi_of(e_g), chi_of(e_c)
        out["chi2_before"] = chi_of(S.ba_linearize(prob, jac=False))
        out["chi2_after_gpu"], out["chi2_after_cpu"] = chi_g, chi_c
        out["chi2_rel_err"] = abs(chi_g - chi_c) / chi_c
    if xp_all is not None:
        cb, x_cpu = cpu_baseline(prob, lam, include_linearize=fused, reps=1)
        xo = x_cpu[:6 * prob["nP"]]
        out["dx_pose_rel_err"] = float(np.abs(xp_all - xo).max() / np.abs(xo).max())
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
