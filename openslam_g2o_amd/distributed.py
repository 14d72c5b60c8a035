"""Multi-GPU sharding of the BlockSolver path: one process per GPU, torch.distributed over
RCCL/xGMI for the single real exchange step.

Partition: rank r owns the contiguous landmark range
[L*r/N, L*(r+1)/N) together with all of their edges, Hll / Dinv / Hpl columns -- with a
pose-local landmark numbering this is a partition by pose-block column ranges.  Assembly
(K2), landmark inversion (K6), the Schur outer products (K7) and back-substitution (K13)
then run on owned data with no communication.  Pose quantities (Hpp diagonal blocks, b_p)
receive contributions from every rank that owns one of the pose's landmarks, and so do the
Hschur blocks near range boundaries; because the Schur complement is linear in those
contributions,

    Hschur = sum_r ( Hpp_r - sum_{lm in r} B Dinv B' ),   bschur = sum_r ( b_p,r - coeff_r ),

one all-reduce(SUM) of the Hschur value array and of bschur is the whole exchange.  Every
rank is given the union block pattern up front (g2ohip_add_schur_pattern) so the arrays
line up element-wise.  LM damping: each rank damps its own landmarks, rank 0 alone adds
lambda to the pose diagonal (g2ohip_set_lambda_split).  The reduced pose system is then
factorised redundantly on every rank (the factorisation does not shard in its reference
form, csparse_helper.cpp:109-140; see DESIGN.md for the subtree-distributed plan) and each
rank back-substitutes its own landmarks, so x_p is replicated and x_l stays sharded.

The local solver is injected (HipBlockSolver in production); tests/test_distributed.py
drives the same code with a CPU stand-in over gloo.
"""
import numpy as np


def landmark_range(n_landmarks, world, rank):
    return (n_landmarks * rank) // world, (n_landmarks * (rank + 1)) // world


def schur_pattern_pairs(pose_idx, lm_idx):
    """Unique (row <= col) pose-block pairs that co-observe a landmark
    (the reduced system's structural blocks, block_solver.hpp:262-288).  Free poses only."""
    pose_idx = np.asarray(pose_idx, np.int64)
    lm_idx = np.asarray(lm_idx, np.int64)
    keep = pose_idx >= 0
    pose_idx, lm_idx = pose_idx[keep], lm_idx[keep]
    if len(pose_idx) == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    order = np.lexsort((pose_idx, lm_idx))
    p, l = pose_idx[order], lm_idx[order]
    n = len(p)
    start = np.flatnonzero(np.r_[True, l[1:] != l[:-1]])
    size = np.diff(np.r_[start, n])
    gid = np.repeat(np.arange(len(start)), size)
    rank_in = np.arange(n) - start[gid]
    cnt = size[gid] - rank_in                       # partners (including itself) at or after each entry
    a = np.repeat(np.arange(n), cnt)
    off = np.arange(len(a)) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    b = a + off
    nmax = int(p.max()) + 1
    keys = np.unique(p[b] * nmax + p[a])            # col * nmax + row, row <= col
    return (keys % nmax).astype(np.int32), (keys // nmax).astype(np.int32)


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f8", data=(int(ptr), False), version=2, strides=None)


def tensor_from_device_ptr(ptr, n, device):
    import torch
    return torch.as_tensor(_DevArray(ptr, n), device=device)


class TorchComm:
    """all-reduce(SUM) through torch.distributed (backend nccl == RCCL on ROCm, gloo on CPU)."""

    def __init__(self, world, force=False):
        self.world = world
        self.force = force    # run the collective even on a 1-rank group (exercises RCCL in single-GPU tests)

    def all_reduce_sum(self, tensors):
        if self.world <= 1 and not self.force:
            return
        import torch.distributed as dist
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)


class ShardedBlockSolver:
    def __init__(self, pose_dim, landmark_dim, rank=0, world=1, device=0, local=None, comm=None, force_exchange=False):
        self.p, self.l = pose_dim, landmark_dim
        self.rank, self.world = rank, world
        self.exchange = world > 1 or force_exchange   # union Schur pattern + all-reduce step active
        if local is None:
            from . import capi
            local = capi.HipBlockSolver(pose_dim, landmark_dim, device)
        self.local = local
        self.comm = comm or TorchComm(world, force=force_exchange)
        self._reduced = None
        self._keep = []

    def parallelism(self):
        return "1 GPU" if self.world == 1 else "landmark-range shards x%d, all-reduce(Hschur,bschur), replicated Cholesky" % self.world

    # ------------------------------------------------------------------------------------
    def setup_ba(self, prob, torch_device=None, nd_leaf=0, fused=False):
        """Shard a bundle-adjustment problem dict (openslam_g2o_amd.synthetic layout: landmark
        index nP + j as vertex 0, pose index as vertex 1) and upload the local edge data.
        fused=True: hand over estimates + measurements instead of Jacobian arrays; errors and Jacobians
        are evaluated inside the assembly kernels (the reference's buildSystem does the same per edge,
        block_solver.hpp:529-532)."""
        nP, nL = prob["nP"], prob["nL"]
        lm0, lm1 = landmark_range(nL, self.world, self.rank)
        lm = prob["v0"].astype(np.int64) - nP
        mine = (lm >= lm0) & (lm < lm1)
        v0 = (nP + (lm[mine] - lm0)).astype(np.int32)
        v1 = prob["v1"][mine].astype(np.int32)
        self.lm0, self.lm1 = lm0, lm1
        self.edge_mask = mine
        if nd_leaf:
            self.local.setOption("nd_leaf", nd_leaf)
        self.set_id = self.local.addEdgeSet(2, v0, v1)
        if self.exchange:
            rows, cols = schur_pattern_pairs(prob["v1"], lm)
            self.local.addSchurPattern(rows, cols)
        self.local.buildStructure(nP, lm1 - lm0, True)
        if fused:
            if torch_device is not None:
                import torch
                self.local.setStream(torch.cuda.current_stream().cuda_stream)
                self._torch_device = torch_device
            else:
                self._torch_device = None
            self.local.baSetEdges(self.set_id, prob["cam_idx"][mine], (prob["pt_idx"][mine] - lm0).astype(np.int32),
                                  prob["meas"][mine], None, prob["f"], prob["cx"], prob["cy"])
            self.local.baSetEstimates(prob["cams"], prob["cam_hidx"], prob["pts"][lm0:lm1], np.arange(lm1 - lm0, dtype=np.int32))
            self.local.baLinearize(True)
            return dict(E_local=int(mine.sum()), L_local=int(lm1 - lm0), lm0=int(lm0), lm1=int(lm1))
        arrays = [np.ascontiguousarray(prob[k][mine]) for k in ("Jp", "Jc", "omega", "err")]
        if torch_device is not None:
            import torch
            dev_arrays = [torch.from_numpy(a).to(torch_device) for a in arrays]
            torch.cuda.synchronize()
            self.local.setStream(torch.cuda.current_stream().cuda_stream)
            self._keep = dev_arrays
            self.local.setEdgeData(self.set_id, *dev_arrays)
            self._torch_device = torch_device
        else:
            self._keep = arrays
            self.local.setEdgeData(self.set_id, *arrays)
            self._torch_device = None
        return dict(E_local=int(mine.sum()), L_local=int(lm1 - lm0), lm0=int(lm0), lm1=int(lm1))

    def _reduced_tensors(self):
        if self._reduced is None:
            if hasattr(self.local, "reducedTensors"):
                self._reduced = self.local.reducedTensors()
            else:
                from . import capi
                ts = []
                for which in (capi.HSCHUR, capi.ARR_BSCHUR):
                    ptr, n = self.local.deviceArray(which)
                    ts.append(tensor_from_device_ptr(ptr, n, self._torch_device))
                self._reduced = ts
        return self._reduced

    # ---- Solver interface (same names as capi.HipBlockSolver) ------------------------------
    def buildSystem(self):
        return self.local.buildSystem()

    def setLambda(self, lam, backup=False):
        return self.local.setLambdaSplit(lam if self.rank == 0 else 0.0, lam, backup)

    def restoreDiagonal(self):
        return self.local.restoreDiagonal()

    def solve(self):
        self.local.solveSchur()
        if self.exchange:
            self.comm.all_reduce_sum(self._reduced_tensors())
        ok = self.local.solveReduced()
        if not ok:
            return False
        self.local.solveBackSubstitute()
        return True

    def chi2(self):
        c = self.local.chi2()
        if self.world > 1:
            import torch
            import torch.distributed as dist
            dev = self._torch_device if self._torch_device is not None else "cpu"
            t = torch.tensor([c], dtype=torch.float64, device=dev)
            dist.all_reduce(t)
            c = float(t.item())
        return c

    def x_poses(self):
        return self.local.x()[:self.p * self.local.nP]

    def x_landmarks_local(self):
        return self.local.x()[self.p * self.local.nP:]
