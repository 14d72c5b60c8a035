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

mode "subtree" (default with HipBlockSolver when world > 1) removes both the replicated
factorisation and the full-size all-reduce.  The elimination-task tree of the reduced system
(nested dissection, sparse_cholesky.hip) is cut near the root into >= N subtrees, dealt to the
ranks; the few tasks above the cut are "shared".  Every pose then has an owner rank (or is
shared), every Hschur block a consumer rank.  Landmarks are dealt to the owner of (the first of)
their observing poses, so nearly all Schur contributions are produced on the rank that consumes
them; only "boundary" blocks (a producer differs from the consumer, or the consumer is shared)
go through a compact gather -> all-reduce -> scatter.  Per solve: boundary blocks + bschur
all-reduce, own subtrees factor/forward (no communication), all-reduce of the subtree roots'
update matrices (tiny: separator-sized), shared top factor/solve + own backward sweep, exchange
of x_p, back-substitution of the own landmarks.  With x_exchange="halo" (default) both vector
exchanges are compact too: b_p only for the poses whose right-hand side has a foreign producer,
x_p only for the foreign poses a rank's landmarks observe -- three latency-sized collectives per
solve; x_p then stays distributed like x_l (every rank holds its own, the shared and its halo
poses; gather_x_poses() assembles the full vector on demand, outside the iteration).
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


def coobservation_pairs(pose_idx, lm_idx):
    """All (row <= col) pose-block pairs co-observing a landmark, one entry per (landmark, pair):
    returns rows, cols, landmark.  Free poses only."""
    pose_idx = np.asarray(pose_idx, np.int64)
    lm_idx = np.asarray(lm_idx, np.int64)
    keep = pose_idx >= 0
    pose_idx, lm_idx = pose_idx[keep], lm_idx[keep]
    order = np.lexsort((pose_idx, lm_idx))
    p, l = pose_idx[order], lm_idx[order]
    n = len(p)
    if n == 0:
        z = np.zeros(0, np.int64)
        return z, z, z
    start = np.flatnonzero(np.r_[True, l[1:] != l[:-1]])
    size = np.diff(np.r_[start, n])
    gid = np.repeat(np.arange(len(start)), size)
    cnt = size[gid] - (np.arange(n) - start[gid])
    a = np.repeat(np.arange(n), cnt)
    b = a + (np.arange(len(a)) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    return p[a], p[b], l[a]


def reduced_pattern(rows, cols, nP):
    """Upper block-CCS (colptr, rowidx, sorted keys col*nP+row) of the reduced system: the given
    pairs plus the full diagonal (block_solver.hpp:262-288)."""
    d = np.arange(nP, dtype=np.int64)
    keys = np.unique(np.r_[np.asarray(cols, np.int64) * nP + np.asarray(rows, np.int64), d * nP + d])
    colptr = np.searchsorted(keys // nP, np.arange(nP + 1)).astype(np.int32)
    return colptr, (keys % nP).astype(np.int32), keys


def assign_landmarks(pose_idx, lm_idx, pose_owner, nL, world):
    """Owner rank of every landmark: the owner of its lowest-numbered observing pose that has one
    (poses of the shared top of the tree have none); unobserved / all-shared landmarks round-robin."""
    pose_idx = np.asarray(pose_idx, np.int64)
    lm_idx = np.asarray(lm_idx, np.int64)
    own = np.where(pose_idx >= 0, pose_owner[np.maximum(pose_idx, 0)], -1)
    ok = own >= 0
    owner = (np.arange(nL) % world).astype(np.int32)
    if ok.any():
        order = np.lexsort((pose_idx[ok], lm_idx[ok]))
        l, o = lm_idx[ok][order], own[ok][order]
        first = np.r_[True, l[1:] != l[:-1]]
        owner[l[first]] = o[first]
    return owner


def boundary_poses(pose_idx, lm_idx, pose_owner, lm_owner):
    """(poses whose b_p is summed over ranks, poses whose x_p another rank needs): an observing
    landmark lives on a rank other than the pose's owner; shared poses (owner -1) only need b_p."""
    pose_idx = np.asarray(pose_idx, np.int64)
    lm_idx = np.asarray(lm_idx, np.int64)
    keep = pose_idx >= 0
    p, l = pose_idx[keep], lm_idx[keep]
    foreign = lm_owner[l] != pose_owner[p]
    bposes = np.unique(p[foreign])
    halo = np.unique(p[foreign & (pose_owner[p] >= 0)])
    return bposes, halo


def boundary_blocks(keys, block_consumer, rows, cols, lms, lm_owner, nP):
    """Indices of the reduced-system blocks whose value must be summed over ranks: a producer (the
    rank owning a landmark co-observed by the block's two poses) differs from the consumer, or the
    consumer is shared."""
    blk = np.searchsorted(keys, cols * nP + rows)
    bad = lm_owner[lms] != block_consumer[blk]
    mark = np.zeros(len(keys), bool)
    mark[blk[bad]] = True
    return np.flatnonzero(mark)


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f8", data=(int(ptr), False), version=2, strides=None)


def tensor_from_device_ptr(ptr, n, device):
    import torch
    t = torch.as_tensor(_DevArray(ptr, n), device=device)
    if t.data_ptr() != int(ptr):   # a silent copy would detach the collective from the solver's memory
        raise RuntimeError("torch made a copy of a device array (wrong device for this pointer?)")
    return t


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
            if t.numel():
                dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def all_reduce_scalar(self, v, device="cpu"):
        if self.world <= 1:
            return v
        import torch
        import torch.distributed as dist
        t = torch.tensor([v], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def all_ok(self, ok, device="cpu"):
        return self.all_reduce_scalar(0.0 if ok else 1.0, device) == 0.0


class HostStagedComm(TorchComm):
    """Same exchange through host memory (gloo): several ranks sharing one GPU (tests), or a box
    without RCCL peer access.  Not a performance path."""

    def all_reduce_sum(self, tensors):
        if self.world <= 1 and not self.force:
            return
        import torch.distributed as dist
        for t in tensors:
            if t.numel():
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.SUM)
                t.copy_(c)


class ShardedBlockSolver:
    def __init__(self, pose_dim, landmark_dim, rank=0, world=1, device=0, local=None, comm=None, force_exchange=False,
                 mode="auto", x_exchange="halo"):
        self.p, self.l = pose_dim, landmark_dim
        self.rank, self.world = rank, world
        self.exchange = world > 1 or force_exchange   # union Schur pattern + all-reduce step active
        if mode == "auto":
            mode = "subtree" if (world > 1 and local is None) else "replicated"
        if mode not in ("subtree", "replicated", "pcg"):
            raise ValueError("mode must be auto, subtree, replicated or pcg")
        self.pcg_tolerance, self.pcg_max_iterations, self.pcg_check_every, self.pcg_iterations = 1e-6, -1, 8, 0
        self.mode = mode
        if x_exchange not in ("halo", "full"):
            raise ValueError("x_exchange must be halo or full")
        self.x_exchange = x_exchange   # subtree mode: exchange only boundary b_p / halo x_p, or the full vectors
        if local is None:
            from . import capi
            local = capi.HipBlockSolver(pose_dim, landmark_dim, device)
        self.local = local
        self.comm = comm or TorchComm(world, force=force_exchange)
        self._reduced = None
        self._keep = []

    def parallelism(self):
        if self.world == 1:
            return "1 GPU"
        if self.mode == "subtree":
            return "x%d: landmarks by pose owner, boundary-block all-reduce, subtree-distributed Cholesky" % self.world
        if self.mode == "pcg":
            return "landmark-range shards x%d, matrix-free PCG, all-reduce(Hschur*d) per iteration" % self.world
        return "landmark-range shards x%d, all-reduce(Hschur,bschur), replicated Cholesky" % self.world

    # ------------------------------------------------------------------------------------
    def setup_ba(self, prob, torch_device=None, nd_leaf=0, fused=False):
        """Shard a bundle-adjustment problem dict (openslam_g2o_amd.synthetic layout: landmark
        index nP + j as vertex 0, pose index as vertex 1) and upload the local edge data.
        fused=True: hand over estimates + measurements instead of Jacobian arrays; errors and Jacobians
        are evaluated inside the assembly kernels (the reference's buildSystem does the same per edge,
        block_solver.hpp:529-532)."""
        nP, nL = prob["nP"], prob["nL"]
        lm = prob["v0"].astype(np.int64) - nP
        if nd_leaf:
            self.local.setOption("nd_leaf", nd_leaf)
        if self.mode == "subtree":
            prow, pcol, plm = coobservation_pairs(prob["v1"], lm)
            colptr, rowidx, keys = reduced_pattern(prow, pcol, nP)
            pose_owner, consumer = self.local.partitionPoses(colptr, rowidx, self.world)
            lm_owner = assign_landmarks(prob["v1"], lm, pose_owner, nL, self.world)
            my = np.flatnonzero(lm_owner == self.rank)
            self.boundary = boundary_blocks(keys, consumer, prow, pcol, plm, lm_owner, nP)
            self.bposes, self.halo = boundary_poses(prob["v1"], lm, pose_owner, lm_owner)
            self.consumer = consumer
            self.pose_owner, self.nnzb_reduced = pose_owner, len(keys)
            rows, cols = (keys % nP).astype(np.int32), (keys // nP).astype(np.int32)
            del prow, pcol, plm
        else:
            lm0, lm1 = landmark_range(nL, self.world, self.rank)
            my = np.arange(lm0, lm1)
            if self.exchange and self.mode != "pcg":
                rows, cols = schur_pattern_pairs(prob["v1"], lm)
        loc = np.full(nL, -1, np.int64)
        loc[my] = np.arange(len(my))
        mine = loc[lm] >= 0
        v0 = (nP + loc[lm[mine]]).astype(np.int32)
        v1 = prob["v1"][mine].astype(np.int32)
        lm0, lm1 = (int(my[0]), int(my[-1]) + 1) if len(my) else (0, 0)
        self.lm0, self.lm1 = lm0, lm1          # (contiguous range in replicated mode)
        self.lm_index = my                      # global landmark id of every local landmark
        self.edge_mask = mine
        self.set_id = self.local.addEdgeSet(2, v0, v1)
        if self.exchange and self.mode != "pcg":   # (the matrix-free mode never forms Hschur: no union pattern needed)
            self.local.addSchurPattern(rows, cols)
        if self.mode == "subtree":
            self.local.setPartition(self.rank, self.world)
            self.local.setOption("mask_solution", 0 if self.x_exchange == "halo" else 1)
        self.local.buildStructure(nP, len(my), True)
        if self.mode == "subtree":
            cp2, ri2 = self.local.pattern(3)
            po2, bc2 = self.local.getPartition()
            if not (np.array_equal(cp2, colptr) and np.array_equal(ri2, rowidx) and np.array_equal(po2, pose_owner)
                    and np.array_equal(bc2, consumer)):
                raise RuntimeError("subtree partition: the solver's reduced pattern/partition differs from the pre-pass")
        if fused:
            if torch_device is not None:
                import torch
                self.local.setStream(torch.cuda.current_stream().cuda_stream)
                self._torch_device = torch_device
            else:
                self._torch_device = None
            if "edge_class" in prob:    # edge classes (several CameraParameters / per-edge robust kernels, g2ohip_ba_set_edges_classes):
                # every rank holds the whole class table, its edges carry their class
                self.local.baSetEdgesClasses(self.set_id, prob["cam_idx"][mine], loc[prob["pt_idx"][mine]].astype(np.int32),
                                             prob["meas"][mine], prob["classes"], prob["edge_class"][mine])
            else:
                # prob["info"]: a per-edge information matrix ([E][4], column-major 2 x 2) read by the kernels; absent:
                # information().setIdentity() declared for the whole set (no per-edge read)
                info = None if prob.get("info") is None else np.ascontiguousarray(prob["info"][mine])
                self.local.baSetEdges(self.set_id, prob["cam_idx"][mine], loc[prob["pt_idx"][mine]].astype(np.int32),
                                      prob["meas"][mine], info, prob["f"], prob["cx"], prob["cy"])
            self.local.baSetEstimates(prob["cams"], prob["cam_hidx"], prob["pts"][my], np.arange(len(my), dtype=np.int32))
            self.local.baLinearize(True)
            return dict(E_local=int(mine.sum()), L_local=int(len(my)), lm0=int(lm0), lm1=int(lm1))
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
        return dict(E_local=int(mine.sum()), L_local=int(len(my)), lm0=int(lm0), lm1=int(lm1))

    def _device_tensor(self, which):
        ptr, n = self.local.deviceArray(which)
        if n == 0:
            import torch
            return torch.zeros(0, dtype=torch.float64, device=self._torch_device)
        return tensor_from_device_ptr(ptr, n, self._torch_device)

    def _subtree_tensors(self):
        if getattr(self, "_sub", None) is None:
            import torch
            from . import capi
            H = self._device_tensor(capi.HSCHUR).view(self.nnzb_reduced, self.p * self.p)
            dev = H.device
            idx = torch.from_numpy(self.boundary).to(dev)
            b = self._device_tensor(capi.ARR_BSCHUR)
            nP = self.local.nP
            x = self._device_tensor(capi.ARR_X)
            sub = dict(H=H, idx=idx, b=b, xbuf=self._device_tensor(capi.ARR_EXCHANGE), xp=self._device_tensor(capi.ARR_XP))
            # after an all-reduce a rank keeps only what it consumes: a block / pose it neither consumes nor produces
            # is not re-formed by its schur_reduce, so a stale sum there would be added again next iteration
            keep = lambda o: torch.from_numpy(((o == self.rank) | (o < 0)).astype(np.float64)).to(dev)[:, None]
            sub.update(hkeep=keep(self.consumer[self.boundary]), bkeep=keep(self.pose_owner[self.bposes]),
                       bkeep_all=keep(self.pose_owner), b2=b.view(nP, self.p))
            if self.x_exchange == "halo":
                nb, p = len(self.boundary), self.p
                sub.update(x2=x[:nP * p].view(nP, p),
                           bidx=torch.from_numpy(self.bposes).to(dev), hidx=torch.from_numpy(self.halo).to(dev),
                           hmine=torch.from_numpy((self.pose_owner[self.halo] == self.rank).astype(np.float64)).to(dev)[:, None],
                           buf1=torch.zeros(nb * p * p + len(self.bposes) * p, dtype=torch.float64, device=dev),
                           buf3=torch.zeros(len(self.halo) * p + 1, dtype=torch.float64, device=dev))
            self._sub = sub
        return self._sub

    def exchange_volume(self):
        """Doubles all-reduced per solve (for DESIGN.md / bench config)."""
        if self.mode != "subtree":
            return sum(int(t.numel()) for t in self._reduced_tensors()) if self.exchange else 0
        t = self._subtree_tensors()
        if self.x_exchange == "halo":
            return int(t["buf1"].numel() + t["xbuf"].numel() + t["buf3"].numel())
        return int(len(self.boundary) * self.p * self.p + t["b"].numel() + t["xbuf"].numel() + t["xp"].numel())

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
        if self.mode == "subtree":   # the solver damps a pose block only on the rank that consumes its diagonal
            return self.local.setLambdaSplit(lam, lam, backup)
        return self.local.setLambdaSplit(lam if self.rank == 0 else 0.0, lam, backup)

    def restoreDiagonal(self):
        return self.local.restoreDiagonal()

    def _native_exchange(self):
        """Pack / unpack kernels and device-side status inside libg2ohip (g2ohip_exchange_*): no torch ops and a single
        synchronisation per solve.  Falls back to the torch formulation for stand-in locals (CPU tests)."""
        if getattr(self, "_native", None) is None:
            self._native = False
            if hasattr(self.local, "exchangeSetup") and self.x_exchange == "halo":
                from . import capi
                keep = lambda o: ((o == self.rank) | (o < 0)).astype(np.float64)
                self.local.exchangeSetup(self.boundary, keep(self.consumer[self.boundary]), self.bposes,
                                         keep(self.pose_owner[self.bposes]), self.halo,
                                         (self.pose_owner[self.halo] == self.rank).astype(np.float64))
                self._nbuf1 = self._device_tensor(capi.ARR_XBOUNDARY)
                self._nbuf3 = self._device_tensor(capi.ARR_XHALO)
                self._nxbuf = self._device_tensor(capi.ARR_EXCHANGE)
                self._native = True
        return self._native

    def _solve_subtree_halo(self):
        """Three latency-sized collectives: boundary Hschur blocks + boundary b_p | subtree roots | halo x_p + status."""
        if self._native_exchange():
            L = self.local
            for _ in range(3):                    # (REPEAT: the status word is a sum over the ranks, every rank loops alike)
                L.solveSchur()
                if len(self.boundary) or len(self.bposes):
                    L.exchangePack(1)
                    self.comm.all_reduce_sum([self._nbuf1])
                    L.exchangeUnpack(1)
                L.solveReducedLocal()
                self.comm.all_reduce_sum([self._nxbuf])
                L.solveReducedShared()
                L.solveReducedFinishAsync()
                L.exchangePack(3)
                self.comm.all_reduce_sum([self._nbuf3])
                L.exchangeUnpack(3)
                L.solveBackSubstitute()           # (harmless after a failed factorisation: the caller discards x)
                ok = L.exchangeStatus()
                if ok is True or ok is False:
                    return ok
            return False
        import torch
        t = self._subtree_tensors()
        p = self.p
        nbb = len(self.boundary) * p * p
        self.local.solveSchur()
        buf1 = t["buf1"]
        if buf1.numel():
            torch.index_select(t["H"], 0, t["idx"], out=buf1[:nbb].view(-1, p * p))
            torch.index_select(t["b2"], 0, t["bidx"], out=buf1[nbb:].view(-1, p))
            self.comm.all_reduce_sum([buf1])
            t["H"].index_copy_(0, t["idx"], buf1[:nbb].view(-1, p * p) * t["hkeep"])
            t["b2"].index_copy_(0, t["bidx"], buf1[nbb:].view(-1, p) * t["bkeep"])
        self.local.solveReducedLocal()
        self.comm.all_reduce_sum([t["xbuf"]])
        self.local.solveReducedShared()
        ok = self.local.solveReducedFinish()
        buf3 = t["buf3"]
        if len(self.halo):
            torch.mul(t["x2"].index_select(0, t["hidx"]), t["hmine"], out=buf3[:-1].view(-1, p))
        buf3[-1] = 0.0 if ok else 1.0
        self.comm.all_reduce_sum([buf3])
        if len(self.halo):
            t["x2"].index_copy_(0, t["hidx"], buf3[:-1].view(-1, p))
        if self.world > 1 or self.comm.force:
            ok = float(buf3[-1].item()) == 0.0
        if not ok:
            return False
        self.local.solveBackSubstitute()
        return True

    def gather_x_poses(self):
        """Full pose increment on every rank (outside the iteration: one all-reduce of 6 P doubles)."""
        import torch
        x = self.x_poses()
        if self.mode != "subtree" or self.x_exchange != "halo" or self.world == 1:
            return x
        own = (self.pose_owner == self.rank) | ((self.pose_owner < 0) & (self.rank == 0))
        t = torch.from_numpy(np.where(np.repeat(own, self.p), x, 0.0))
        dev = self._torch_device if getattr(self, "_torch_device", None) is not None else "cpu"
        t = t.to(dev)
        self.comm.all_reduce_sum([t])
        return t.cpu().numpy()

    def _solve_subtree(self):
        if self.x_exchange == "halo":
            return self._solve_subtree_halo()
        t = self._subtree_tensors()
        self.local.solveSchur()
        if len(self.boundary):
            buf = t["H"].index_select(0, t["idx"])
            self.comm.all_reduce_sum([buf])
            t["H"].index_copy_(0, t["idx"], buf * t["hkeep"])
        self.comm.all_reduce_sum([t["b"]])
        t["b2"].mul_(t["bkeep_all"])
        self.local.solveReducedLocal()          # own subtrees: factor + forward sweep, pack the roots
        self.comm.all_reduce_sum([t["xbuf"]])
        self.local.solveReducedShared()         # shared top, then back down the own subtrees; x_p masked
        self.comm.all_reduce_sum([t["xp"]])
        ok = self.local.solveReducedFinish()
        ok = self.comm.all_ok(ok, t["b"].device)
        if not ok:
            return False
        self.local.solveBackSubstitute()
        return True

    def _solve_pcg(self):
        """LinearSolverPCG's iteration (linear_solver_pcg.hpp:79-196, block-Jacobi, relative tolerance) on the reduced system
        with landmarks sharded over the ranks and Hschur never formed: every rank applies ITS summand
        (Hpp_r + lambda_r I - Hpl_r Dinv_r Hpl_r') d (g2ohip_schur_operator_apply), one all-reduce per iteration makes it the
        full product; bschur and the preconditioner blocks are summed once per solve.  Vectors are torch tensors on the
        solver's stream; the scalars stay on the device, the host looks at the residual every pcg_check_every iterations."""
        import torch
        from . import capi
        L, p, nP = self.local, self.p, self.local.nP
        L.schurOperatorPrepare()
        b = self._device_tensor(capi.ARR_BSCHUR)
        D = self._device_tensor(capi.ARR_SCHUR_DIAG)
        self.comm.all_reduce_sum([b, D])
        J = torch.linalg.inv(D.view(nP, p, p).transpose(1, 2)).contiguous()      # (blocks are stored column-major)
        x = self._device_tensor(capi.ARR_X)[:nP * p]
        x.zero_()
        r = b.clone()
        d = torch.bmm(J, r.view(nP, p, 1)).view(-1)
        q = torch.empty_like(d)
        dn = torch.dot(r, d)
        d0 = float(self.pcg_tolerance * dn.item())
        max_iter = self.pcg_max_iterations if self.pcg_max_iterations > 0 else nP * p
        it, ok = 0, True
        while it < max_iter:
            L.schurOperatorApply(d.data_ptr(), q.data_ptr())
            self.comm.all_reduce_sum([q])
            alpha = dn / torch.dot(d, q)
            x.add_(alpha * d)
            r.sub_(alpha * q)
            s = torch.bmm(J, r.view(nP, p, 1)).view(-1)
            dold = dn
            dn = torch.dot(r, s)
            d = s + (dn / dold) * d
            it += 1
            if it % self.pcg_check_every == 0 or it == max_iter:
                v = float(dn.item())
                if not np.isfinite(v):
                    ok = False
                    break
                if v <= d0:
                    break
        self.pcg_iterations = it
        if not ok:
            return False
        L.solveBackSubstitute()
        return True

    # ---- collectives inside libg2ohip (g2ohip_comm_*, g2ohip_solve_sharded): no torch op in the iteration ----------
    def attach_library_comm(self, kind="auto"):
        """Give the local solver its own communicator and let the library run the whole sharded solve and the LM scalars.
        kind "rccl": ncclCommInitRank inside the library (the unique id travels through torch.distributed once);
        "host": an all-reduce over host memory through torch.distributed (gloo) -- ranks sharing one GPU, tests;
        "peer": mailboxes in device memory exported with hipIpc handles and written by the peers directly (opt-in; the handles
        and the host scalars travel through torch.distributed);
        "auto": rccl when the process group's backend is nccl, host otherwise.  Returns True when attached."""
        if self.mode != "subtree" or self.x_exchange != "halo" or not hasattr(self.local, "solveSharded"):
            return False
        if not self._native_exchange():
            return False
        import torch
        import torch.distributed as dist
        have_pg = dist.is_available() and dist.is_initialized()
        if kind == "auto":
            kind = "rccl" if (have_pg and dist.get_backend() == "nccl") or (self.world == 1 and not have_pg) else "host"
        if kind == "rccl":
            if self.world > 1:
                idt = torch.zeros(128, dtype=torch.uint8, device=self._torch_device)
                if self.rank == 0:
                    idt = torch.frombuffer(bytearray(self.local.commUniqueId()), dtype=torch.uint8).to(self._torch_device)
                dist.broadcast(idt, src=0)
                uid = bytes(idt.cpu().numpy().tobytes())
            else:
                uid = self.local.commUniqueId()
            self.local.commInitRccl(self.rank, self.world, uid)
        else:
            host_group = None
            if have_pg and self.world > 1 and dist.get_backend() == "nccl":
                host_group = dist.new_group(backend="gloo")     # (host memory does not travel through RCCL)

            def host_all_reduce(buf, op):
                if self.world <= 1:
                    return
                t = torch.from_numpy(buf)          # shares the pinned staging memory: reduced in place
                dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=host_group)
            if kind == "peer":
                self.local.commInitPeer(self.rank, self.world, host_all_reduce)
            else:
                self.local.commInitHost(self.rank, self.world, host_all_reduce)
        self._lib_comm = kind
        return True

    def setRobustKernel(self, kind, delta):
        return self.local.setRobustKernel(self.set_id, kind, delta)

    def maxDiagonal(self):
        if getattr(self, "_lib_comm", None):
            return self.local.maxDiagonalSharded()
        if self.world <= 1:
            return self.local.maxDiagonal()
        raise RuntimeError("maxDiagonal over several ranks needs attach_library_comm()")

    def computeScale(self, lam):
        if getattr(self, "_lib_comm", None):
            return self.local.computeScaleSharded(lam)
        if self.world <= 1:
            return self.local.computeScale(lam)
        raise RuntimeError("computeScale over several ranks needs attach_library_comm()")

    def solve(self):
        if self.mode == "pcg":
            return self._solve_pcg()
        if self.mode == "subtree" and getattr(self, "_lib_comm", None):
            return self.local.solveSharded()
        if self.mode == "subtree":
            return self._solve_subtree()
        if not self.exchange:
            return self.local.solve()     # one rank: the solver may fold the Schur reduction into the factorisation
        self.local.solveSchur()
        self.comm.all_reduce_sum(self._reduced_tensors())
        ok = self.local.solveReduced()
        if not ok:
            return False
        self.local.solveBackSubstitute()
        return True

    def chi2(self):
        if getattr(self, "_lib_comm", None):
            return self.local.chi2Sharded()
        dev = self._torch_device if getattr(self, "_torch_device", None) is not None else "cpu"
        return self.comm.all_reduce_scalar(self.local.chi2(), dev)

    def x_poses(self):
        return self.local.x()[:self.p * self.local.nP]

    def x_landmarks_local(self):
        return self.local.x()[self.p * self.local.nP:]


class ShardedBAGraph:
    """lm.py's graph protocol over a ShardedBlockSolver whose shard was set up with fused=True: every rank linearises,
    updates and stacks ITS edges / landmarks and its copy of the cameras (x_p is valid for the cameras its landmarks
    observe: own, shared, halo); chi2 is the sum over the ranks (sparse_optimizer.cpp:100-114)."""

    device_resident = False     # (the fused solve_async / trial_stats pair is a single-GPU shortcut)

    def __init__(self, solver):
        self.s = solver

    def linearize(self):
        self.s.local.baLinearize(True)

    def compute_active_errors(self):
        self.s.local.baLinearize(False)

    def chi2(self):
        return self.s.chi2()

    def update(self):
        self.s.local.baUpdate()

    def push(self):
        self.s.local.baPush()

    def pop(self):
        self.s.local.baPop()

    def discard_top(self):
        self.s.local.baDiscardTop()

