#!/usr/bin/env python
"""Latency of the library's all-reduce for the payload of a sharded solve (94 KB = 12 032 doubles), peer mailboxes against the host
staging, N processes sharing ONE GPU (what this box offers: launch + flag latency, no xGMI hop):
python tools/probe/peer_allreduce.py [world=4] [doubles=12032]"""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(rank, world, port, n, kind):
    import numpy as np, torch, torch.distributed as dist
    from openslam_g2o_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    s = capi.HipBlockSolver(6, 3, 0)

    def host_all_reduce(buf, op):
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
    if kind == "peer":
        s.commInitPeer(rank, world, host_all_reduce)
    else:
        s.commInitHost(rank, world, host_all_reduce)
    x = torch.full((n,), float(rank + 1), dtype=torch.float64, device="cuda")
    s.commAllReduce(x.data_ptr(), n); torch.cuda.synchronize()
    assert float(x[0]) == world * (world + 1) / 2 and float(x[-1]) == float(x[0])
    reps = 300
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        x.fill_(1.0)
        s.commAllReduce(x.data_ptr(), n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert float(x[0]) == world
    if rank == 0:
        print("%-5s world %d  %6d doubles  %8.1f us per all-reduce" % (kind, world, n, 1e6 * dt), flush=True)
    dist.barrier()
    s.commDestroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12032
    for kind in ("host", "peer"):
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        mp.spawn(worker, args=(world, port, n, kind), nprocs=world, join=True)
