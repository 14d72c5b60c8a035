#!/usr/bin/env python
"""Secondary measurement (not the headline metric): whole Levenberg-Marquardt iterations of BASELINE.json's
config 5 -- 100k-pose / 1M-landmark BA, Huber kernel (delta = 1), 5 % outliers, tau = 1e-5, <= 10 trials --
with estimates, errors and Jacobians resident on the device.  Prints one JSON line on rank 0.

    python lm_bench.py                                  one GPU
    python lm_bench.py --gpus N                         N GPUs (re-executes itself under torch.distributed.run; also
                                                        accepts being started by a launcher): sharded buildSystem / Schur / subtree-distributed Cholesky,
                                                        collectives inside libg2ohip (RCCL), one process per GPU
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--poses", type=int, default=100000)
    ap.add_argument("--landmarks", type=int, default=1000000)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--comm", choices=["rccl", "staged"], default="rccl",
                    help="staged: gloo + host staging with every rank on cuda:0 (functional check on a 1-GPU box)")
    args = ap.parse_args()
    from openslam_g2o_amd.launch import relaunch_if_needed
    relaunch_if_needed(args.gpus, __file__)
    from openslam_g2o_amd import capi, lm, synthetic as S
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    prob = S.make_ba_problem(args.poses, args.landmarks, outlier_frac=0.05)
    sync = None
    if world == 1:
        s, g = lm.setup_device_ba(prob, huber_delta=1.0)
        sync = s.sync
        barrier = lambda: s.sync()
        par = "1 GPU"
    else:
        import torch
        import torch.distributed as dist
        from openslam_g2o_amd import distributed as D
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.comm == "rccl" and torch.cuda.device_count() < world:
            args.comm = "staged"      # a 1-GPU box: RCCL refuses two ranks per device
        if args.comm == "staged":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("gloo" if args.comm == "staged" else "nccl", **({} if args.comm == "staged" else {"device_id": dev}))
        s = D.ShardedBlockSolver(6, 3, rank=rank, world=world, device=local_rank,
                                 comm=D.HostStagedComm(world) if args.comm == "staged" else None, mode="subtree")
        side = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(side)
        s.setup_ba(prob, torch_device=dev, fused=True)
        if not s.attach_library_comm("host" if args.comm == "staged" else "rccl"):
            raise SystemExit("the library communicator could not be attached")
        s.setRobustKernel(capi.KERNEL_HUBER, 1.0)
        g = D.ShardedBAGraph(s)

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
        par = s.parallelism()
    g.compute_active_errors()
    chi0 = g.chi2()
    barrier()
    t0 = time.perf_counter()
    it_s = []
    n, chis, lams, trials = lm.optimize(g, s, args.iterations, "lm", times=it_s)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.comm == "staged" else torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"workload": "config 5: %d poses / %d landmarks / %d observations, Huber delta=1, 5%% outliers, LM tau=1e-5" % (
            args.poses, args.landmarks, prob["E"]), "n_gpus": world, "parallelism": par, "iterations": n, "lm_trials": trials,
            "ms_per_lm_iteration": 1e3 * dt / max(n, 1), "ms_per_lm_trial": 1e3 * dt / max(sum(trials), 1),
            "ms_iterations": [round(1e3 * v, 4) for v in it_s],     # (the first carries one-time costs: code loading, graph capture)
            "ms_per_lm_iteration_steady": 1e3 * sorted(it_s[1:])[len(it_s[1:]) // 2] if len(it_s) > 2 else None, "chi2_initial": chi0,
            "chi2": chis, "lambda": lams}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
