"""ctypes binding of include/g2ohip.h and the host-side mirror of g2o's Solver interface.

`HipBlockSolver` keeps the method names, argument meaning and error behaviour of
g2o::Solver / g2o::BlockSolver<Traits> (/root/reference/g2o/core/solver.h:44-149,
block_solver.h:98-178): init, buildStructure, buildSystem, solve (-> bool), setLambda,
restoreDiagonal, x, b, vectorSize ...  `HipLinearSolver` mirrors g2o::LinearSolver<M>
(/root/reference/g2o/core/linear_solver.h:40-81).

The HIP library is the only compute path: if libg2ohip.so is missing or no GPU is visible
the calls raise -- there is no CPU fallback.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# G2OHIP_LIB: an instrumented build of the same sources (make VARIANT=...), for the profiling tools only
LIB_PATH = os.environ.get("G2OHIP_LIB") or os.path.join(_HERE, "lib", "libg2ohip.so")

c_int_p = C.POINTER(C.c_int32)
c_dbl_p = C.POINTER(C.c_double)

OK, NOT_PD, REPEAT = 0, 1, 2
ERR_UNSUPPORTED = -4
HPP, HPL, HLL, HSCHUR, DINV = 0, 1, 2, 3, 4
PART_NO_VERTEX0, PART_NO_VERTEX1, PART_NO_CHI2 = 1, 2, 4      # g2ohip_set_edge_set_parts
ARR_BSCHUR, ARR_X, ARR_B, ARR_EXCHANGE, ARR_XP, ARR_XBOUNDARY, ARR_XHALO, ARR_SCHUR_DIAG = 100, 101, 102, 103, 104, 105, 106, 107
KERNEL_NONE, KERNEL_HUBER, KERNEL_PSEUDOHUBER, KERNEL_CAUCHY, KERNEL_SATURATED, KERNEL_DCS = 0, 1, 2, 3, 4, 5


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "timeQuadraticForm", "timeSchurComplement", "timeSymbolicDecomposition", "timeNumericDecomposition",
        "timeLinearSolution", "timeLinearSolver", "timeBackSubstitution")] + [(n, C.c_size_t) for n in (
            "hessianDimension", "hessianPoseDimension", "hessianLandmarkDimension", "choleskyNNZ", "numFronts",
            "numLevels", "maxFrontDim", "iterationsLinearSolver")] + [(n, C.c_double) for n in (
                "timeResiduals", "timeLinearize", "timeUpdate")] + [("dependencyFallbacks", C.c_size_t), ("bandChains", C.c_size_t), ("bandCholeskyNNZ", C.c_size_t), ("bandPivots", C.c_size_t), ("shardedCollectives", C.c_size_t), ("treeBackwardGroups", C.c_size_t), ("choleskyFlops", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


EXPORTS = [
    "g2ohip_last_error", "g2ohip_device_count", "g2ohip_create", "g2ohip_destroy", "g2ohip_set_stream", "g2ohip_init",
    "g2ohip_add_edge_set", "g2ohip_build_structure", "g2ohip_set_edge_data", "g2ohip_set_edge_errors", "g2ohip_set_robust_kernel", "g2ohip_set_robust_kernel_per_edge",
    "g2ohip_build_system", "g2ohip_chi2", "g2ohip_set_lambda", "g2ohip_restore_diagonal", "g2ohip_max_diagonal",
    "g2ohip_compute_scale", "g2ohip_solve", "g2ohip_vector_size", "g2ohip_copy_x", "g2ohip_copy_b", "g2ohip_x_device",
    "g2ohip_b_device", "g2ohip_multiply_hessian", "g2ohip_sync", "g2ohip_set_profiling", "g2ohip_get_stats",
    "g2ohip_set_option", "g2ohip_get_nnzb", "g2ohip_get_pattern", "g2ohip_copy_values", "g2ohip_device_array",
    "g2ohip_solve_schur", "g2ohip_solve_reduced", "g2ohip_solve_back_substitute", "g2ohip_ls_create",
    "g2ohip_ls_destroy", "g2ohip_ls_init", "g2ohip_ls_solve", "g2ohip_ls_solve_pattern", "g2ohip_ls_get_stats", "g2ohip_ls_set_option",
    "g2ohip_kernel_slots", "g2ohip_kernel_name", "g2ohip_kernel_time", "g2ohip_add_schur_pattern", "g2ohip_set_edge_set_parts",
    "g2ohip_set_lambda_split", "g2ohip_host_register", "g2ohip_host_unregister", "g2ohip_ba_set_edges", "g2ohip_ba_set_edges_classes", "g2ohip_ba_set_estimates", "g2ohip_ba_get_estimates", "g2ohip_ba_get_estimates_of", "g2ohip_ba_fetch_estimates_begin", "g2ohip_ba_fetch_estimates_wait",
    "g2ohip_ba_linearize", "g2ohip_ba_update", "g2ohip_ba_push", "g2ohip_ba_pop", "g2ohip_ba_discard_top",
    "g2ohip_set_partition", "g2ohip_solve_reduced_local", "g2ohip_solve_reduced_shared", "g2ohip_solve_reduced_finish",
    "g2ohip_schur_operator_prepare", "g2ohip_schur_operator_apply", "g2ohip_solve_async", "g2ohip_trial_stats_begin", "g2ohip_trial_stats", "g2ohip_solve_reduced_finish_async", "g2ohip_exchange_setup", "g2ohip_exchange_pack", "g2ohip_exchange_unpack", "g2ohip_exchange_status",
    "g2ohip_get_partition", "g2ohip_partition_poses",
    "g2ohip_pg_set_edges", "g2ohip_pg_set_estimates", "g2ohip_pg_get_estimates", "g2ohip_pg_linearize", "g2ohip_pg_update",
    "g2ohip_pg_push", "g2ohip_pg_pop", "g2ohip_pg_discard_top", "g2ohip_copy_edge_data",
    "g2ohip_compute_marginals", "g2ohip_set_x", "g2ohip_copy_diagonal",
    "g2ohip_comm_unique_id", "g2ohip_comm_init_rccl", "g2ohip_comm_init_host", "g2ohip_comm_init_peer", "g2ohip_comm_destroy", "g2ohip_comm_all_reduce",
    "g2ohip_update_structure", "g2ohip_clear_edge_sets", "g2ohip_solve_sharded", "g2ohip_chi2_sharded", "g2ohip_max_diagonal_sharded", "g2ohip_compute_scale_sharded",
]

HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_int)

_lib = None


class G2oHipError(RuntimeError):
    pass


def load():
    """Load libg2ohip.so (raises if it was not built: build with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise G2oHipError("libg2ohip.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`"
                          % LIB_PATH)
    # One HIP/HSA runtime per process: PyTorch ships its own libamdhip64 / libhsa-runtime64 (same SONAMEs as
    # /opt/rocm's).  If this library pulled in the system copies first, a later `import torch` would bring up a
    # second HSA runtime that finds no GPU; importing torch first makes its copies the ones both sides use.
    # (Plain C/C++ consumers of the ABI link the system runtime and are not affected.)
    if "torch" not in sys.modules and os.environ.get("G2OHIP_NO_TORCH_PRELOAD") is None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.g2ohip_last_error.restype = C.c_char_p
    L.g2ohip_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int]
    L.g2ohip_destroy.argtypes = [vp]
    L.g2ohip_destroy.restype = None
    L.g2ohip_set_stream.argtypes = [vp, vp]
    L.g2ohip_init.argtypes = [vp]
    L.g2ohip_add_edge_set.argtypes = [vp, C.c_int, C.c_int, c_int_p, c_int_p]
    L.g2ohip_build_structure.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.g2ohip_set_edge_data.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int]
    L.g2ohip_set_robust_kernel.argtypes = [vp, C.c_int, C.c_int, C.c_double]
    L.g2ohip_set_robust_kernel_per_edge.argtypes = [vp, C.c_int, c_int_p, c_dbl_p]
    L.g2ohip_build_system.argtypes = [vp]
    L.g2ohip_chi2.argtypes = [vp, c_dbl_p]
    L.g2ohip_set_lambda.argtypes = [vp, C.c_double, C.c_int]
    L.g2ohip_restore_diagonal.argtypes = [vp]
    L.g2ohip_max_diagonal.argtypes = [vp, c_dbl_p]
    L.g2ohip_compute_scale.argtypes = [vp, C.c_double, c_dbl_p]
    L.g2ohip_set_partition.argtypes = [vp, C.c_int, C.c_int]
    L.g2ohip_get_partition.argtypes = [vp, c_int_p, c_int_p]
    L.g2ohip_partition_poses.argtypes = [vp, C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p]
    for n in ("g2ohip_solve", "g2ohip_solve_schur", "g2ohip_solve_reduced", "g2ohip_solve_back_substitute",
              "g2ohip_sync", "g2ohip_solve_reduced_local", "g2ohip_solve_reduced_shared",
              "g2ohip_solve_reduced_finish", "g2ohip_solve_reduced_finish_async", "g2ohip_exchange_status", "g2ohip_solve_async"):
        getattr(L, n).argtypes = [vp]
    L.g2ohip_exchange_setup.argtypes = [vp, C.c_int, c_int_p, c_dbl_p, C.c_int, c_int_p, c_dbl_p, C.c_int, c_int_p, c_dbl_p]
    L.g2ohip_trial_stats.argtypes = [vp, C.c_double, c_int_p, c_dbl_p, c_dbl_p]
    L.g2ohip_trial_stats_begin.argtypes = [vp, C.c_double]
    L.g2ohip_schur_operator_prepare.argtypes = [vp]
    L.g2ohip_schur_operator_apply.argtypes = [vp, C.c_void_p, C.c_void_p]
    L.g2ohip_exchange_pack.argtypes = [vp, C.c_int]
    L.g2ohip_exchange_unpack.argtypes = [vp, C.c_int]
    L.g2ohip_vector_size.argtypes = [vp]
    L.g2ohip_vector_size.restype = C.c_size_t
    L.g2ohip_copy_x.argtypes = [vp, c_dbl_p]
    L.g2ohip_copy_b.argtypes = [vp, c_dbl_p]
    L.g2ohip_set_x.argtypes = [vp, c_dbl_p]
    L.g2ohip_x_device.argtypes = [vp]
    L.g2ohip_x_device.restype = vp
    L.g2ohip_b_device.argtypes = [vp]
    L.g2ohip_b_device.restype = vp
    L.g2ohip_multiply_hessian.argtypes = [vp, c_dbl_p, c_dbl_p]
    L.g2ohip_set_profiling.argtypes = [vp, C.c_int]
    L.g2ohip_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.g2ohip_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    L.g2ohip_get_nnzb.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.g2ohip_get_pattern.argtypes = [vp, C.c_int, c_int_p, c_int_p]
    L.g2ohip_copy_values.argtypes = [vp, C.c_int, c_dbl_p]
    L.g2ohip_device_array.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.g2ohip_add_schur_pattern.argtypes = [vp, C.c_int, c_int_p, c_int_p]
    L.g2ohip_set_edge_set_parts.argtypes = [vp, C.c_int, C.c_int]
    L.g2ohip_set_lambda_split.argtypes = [vp, C.c_double, C.c_double, C.c_int]
    L.g2ohip_ba_set_edges.argtypes = [vp, C.c_int, c_int_p, c_int_p, c_dbl_p, c_dbl_p, C.c_double, C.c_double, C.c_double]
    L.g2ohip_ba_set_edges_classes.argtypes = [vp, C.c_int, c_int_p, c_int_p, c_dbl_p, c_dbl_p, C.c_int, c_dbl_p, c_int_p]
    L.g2ohip_ba_set_estimates.argtypes = [vp, C.c_int, c_dbl_p, c_int_p, C.c_int, c_dbl_p, c_int_p]
    L.g2ohip_ba_get_estimates.argtypes = [vp, c_dbl_p, c_dbl_p]
    L.g2ohip_ba_get_estimates_of.argtypes = [vp, C.c_int, c_int_p, c_dbl_p, C.c_int, c_int_p, c_dbl_p]
    L.g2ohip_set_edge_errors.argtypes = [vp, C.c_int, c_dbl_p]
    L.g2ohip_ba_fetch_estimates_begin.argtypes = [vp, c_dbl_p, c_dbl_p, C.c_int]
    L.g2ohip_ba_fetch_estimates_wait.argtypes = [vp, C.c_int]
    L.g2ohip_ba_linearize.argtypes = [vp, C.c_int]
    for n in ("g2ohip_ba_update", "g2ohip_ba_push", "g2ohip_ba_pop", "g2ohip_ba_discard_top", "g2ohip_pg_update", "g2ohip_pg_push",
              "g2ohip_pg_pop", "g2ohip_pg_discard_top"):
        getattr(L, n).argtypes = [vp]
    L.g2ohip_pg_set_edges.argtypes = [vp, C.c_int, C.c_int, c_int_p, c_int_p, c_dbl_p, c_dbl_p]
    L.g2ohip_pg_set_estimates.argtypes = [vp, C.c_int, c_dbl_p, c_int_p]
    L.g2ohip_pg_get_estimates.argtypes = [vp, c_dbl_p]
    L.g2ohip_pg_linearize.argtypes = [vp, C.c_int]
    L.g2ohip_copy_edge_data.argtypes = [vp, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p]
    L.g2ohip_compute_marginals.argtypes = [vp, C.c_int, c_int_p, c_int_p, c_dbl_p]
    L.g2ohip_copy_diagonal.argtypes = [vp, c_dbl_p]
    L.g2ohip_comm_unique_id.argtypes = [C.c_char_p]
    L.g2ohip_comm_init_rccl.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.g2ohip_comm_init_host.argtypes = [vp, C.c_int, C.c_int, HOST_ALLREDUCE_FN, vp]
    L.g2ohip_comm_init_peer.argtypes = [vp, C.c_int, C.c_int, HOST_ALLREDUCE_FN, vp, C.c_size_t]
    L.g2ohip_comm_destroy.argtypes = [vp]
    L.g2ohip_comm_all_reduce.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.g2ohip_update_structure.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.g2ohip_solve_sharded.argtypes = [vp]
    L.g2ohip_chi2_sharded.argtypes = [vp, c_dbl_p]
    L.g2ohip_max_diagonal_sharded.argtypes = [vp, c_dbl_p]
    L.g2ohip_compute_scale_sharded.argtypes = [vp, C.c_double, c_dbl_p]
    L.g2ohip_kernel_name.argtypes = [C.c_int]
    L.g2ohip_kernel_name.restype = C.c_char_p
    L.g2ohip_kernel_time.argtypes = [vp, C.c_int, c_dbl_p, C.POINTER(C.c_long), C.c_int]
    L.g2ohip_ls_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.g2ohip_ls_destroy.argtypes = [vp]
    L.g2ohip_ls_destroy.restype = None
    L.g2ohip_ls_init.argtypes = [vp]
    L.g2ohip_ls_solve.argtypes = [vp, C.c_int, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p]
    L.g2ohip_ls_solve_pattern.argtypes = [vp, C.c_int, c_int_p, c_int_p, c_dbl_p, C.c_int, c_int_p, c_int_p, c_dbl_p]
    L.g2ohip_ls_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.g2ohip_ls_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    _lib = L
    return L


def _check(rc, what):
    if rc < 0:
        raise G2oHipError("%s failed (%d): %s" % (what, rc, load().g2ohip_last_error().decode()))
    return rc


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int_p)


def _dp(a):
    return a.ctypes.data_as(c_dbl_p)


def _ptr(a):
    """Host numpy array, torch CUDA tensor or raw int -> (void*, on_device)."""
    if a is None:
        return None, None
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data), False
    if isinstance(a, int):
        return C.c_void_p(a), True
    if hasattr(a, "data_ptr"):  # torch tensor
        return C.c_void_p(a.data_ptr()), bool(a.is_cuda)
    raise TypeError(type(a))


class HipBlockSolver:
    """g2o::BlockSolver<BlockSolverTraits<p,l>> on one MI355X (block_solver.h:98-178)."""

    def __init__(self, pose_dim, landmark_dim, device=0):
        self.L = load()
        self.p, self.l = pose_dim, landmark_dim
        self.h = C.c_void_p()
        _check(self.L.g2ohip_create(C.byref(self.h), pose_dim, landmark_dim, device), "g2ohip_create")
        self._keep = {}
        self.nP = self.nL = 0
        # (G2OHIP_OPTIONS="name=value,..." in the environment is applied by g2ohip_create itself)

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.g2ohip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Solver interface ------------------------------------------------------------
    def init(self):
        _check(self.L.g2ohip_init(self.h), "init")
        return True

    def addEdgeSet(self, error_dim, v0, v1=None):
        v0 = _i32(v0)
        v1 = None if v1 is None else _i32(v1)
        if v1 is not None and len(v1) != len(v0):
            raise ValueError("addEdgeSet: v0 and v1 must have one entry per edge")
        sid = _check(self.L.g2ohip_add_edge_set(self.h, error_dim, len(v0), _ip(v0), _ip(v1)), "add_edge_set")
        if not hasattr(self, "_set_sizes"):
            self._set_sizes = {}
        self._set_sizes[sid] = len(v0)      # (the wrappers check array lengths against it: the C ABI takes bare pointers)
        return sid

    def buildStructure(self, num_poses, num_landmarks=0, schur=None):
        if schur is None:
            schur = num_landmarks > 0
        _check(self.L.g2ohip_build_structure(self.h, num_poses, num_landmarks, int(bool(schur))), "buildStructure")
        self.nP, self.nL = num_poses, num_landmarks
        self.schur = bool(schur)
        return True

    def setEdgeData(self, set_id, J0, J1, omega, err):
        """Host numpy arrays (copied) or torch CUDA tensors (zero-copy, must outlive buildSystem)."""
        arrs = []
        dev = None
        for a in (J0, J1, omega, err):
            if isinstance(a, (list, tuple)):
                a = np.asarray(a)
            if isinstance(a, np.ndarray):
                a = _f64(a)
            arrs.append(a)
        ptrs = []
        for a in arrs:
            p, d = _ptr(a)
            ptrs.append(p)
            if d is not None:
                assert dev is None or dev == d, "mixing host and device arrays"
                dev = d
        n = getattr(self, "_set_sizes", {}).get(set_id)
        if n:      # the C ABI takes bare pointers: a short array would be read out of bounds during the upload
            numel = lambda a: int(a.size) if isinstance(a, np.ndarray) else int(a.numel())
            ne, no = numel(arrs[3]), numel(arrs[2])
            d = ne // n
            if d * n != ne or no != n * d * d or numel(arrs[0]) % (n * d) or (arrs[1] is not None and numel(arrs[1]) % (n * d)):
                raise ValueError("setEdgeData: array sizes do not fit the %d edges of set %d" % (n, set_id))
        self._keep[set_id] = arrs
        _check(self.L.g2ohip_set_edge_data(self.h, set_id, ptrs[0], ptrs[1], ptrs[2], ptrs[3], int(bool(dev))),
               "setEdgeData")

    def setRobustKernel(self, set_id, kind, delta=1.0):
        _check(self.L.g2ohip_set_robust_kernel(self.h, set_id, kind, delta), "setRobustKernel")

    def setRobustKernelPerEdge(self, set_id, kinds, deltas):
        """One robust kernel per edge (kinds [n] as KERNEL_*, deltas [n]); kinds=None returns to the set-level kernel."""
        if kinds is None:
            _check(self.L.g2ohip_set_robust_kernel_per_edge(self.h, set_id, None, None), "setRobustKernelPerEdge")
            return
        k, d = _i32(kinds), _f64(deltas)
        if len(k) != self._set_sizes[set_id] or len(d) != len(k):
            raise ValueError("setRobustKernelPerEdge: one entry per edge of set %d" % set_id)
        _check(self.L.g2ohip_set_robust_kernel_per_edge(self.h, set_id, _ip(k), _dp(d)), "setRobustKernelPerEdge")

    def buildSystem(self):
        _check(self.L.g2ohip_build_system(self.h), "buildSystem")
        return True

    def chi2(self):
        v = C.c_double()
        _check(self.L.g2ohip_chi2(self.h, C.byref(v)), "chi2")
        return v.value

    def setLambda(self, lam, backup=False):
        _check(self.L.g2ohip_set_lambda(self.h, lam, int(backup)), "setLambda")
        return True

    def setLambdaSplit(self, lam_pose, lam_landmark, backup=False):
        _check(self.L.g2ohip_set_lambda_split(self.h, lam_pose, lam_landmark, int(backup)), "setLambdaSplit")
        return True

    def setEdgeSetParts(self, set_id, parts):
        """One binary set per vertex pair of n-ary edges (BaseMultiEdge): PART_NO_VERTEX0 / PART_NO_VERTEX1 / PART_NO_CHI2."""
        _check(self.L.g2ohip_set_edge_set_parts(self.h, set_id, int(parts)), "setEdgeSetParts")

    def addSchurPattern(self, rows, cols):
        rows, cols = _i32(rows), _i32(cols)
        _check(self.L.g2ohip_add_schur_pattern(self.h, len(rows), _ip(rows), _ip(cols)), "addSchurPattern")

    def restoreDiagonal(self):
        _check(self.L.g2ohip_restore_diagonal(self.h), "restoreDiagonal")

    def maxDiagonal(self):
        v = C.c_double()
        _check(self.L.g2ohip_max_diagonal(self.h, C.byref(v)), "maxDiagonal")
        return v.value

    def computeScale(self, lam):
        v = C.c_double()
        _check(self.L.g2ohip_compute_scale(self.h, lam, C.byref(v)), "computeScale")
        return v.value

    def solve(self):
        """True on success, False when the system is not positive definite (like the reference)."""
        return _check(self.L.g2ohip_solve(self.h), "solve") == OK

    def solveSchur(self):
        _check(self.L.g2ohip_solve_schur(self.h), "solveSchur")

    def solveReduced(self):
        return _check(self.L.g2ohip_solve_reduced(self.h), "solveReduced") == OK

    # ---- subtree-distributed reduced solve (include/g2ohip.h, "Subtree-distributed factorisation")
    def setPartition(self, rank, world):
        _check(self.L.g2ohip_set_partition(self.h, rank, world), "setPartition")

    def getPartition(self):
        """(pose_owner[nP], block_consumer[nnzb of the reduced system]); -1 = shared by all ranks."""
        po = np.zeros(self.nP, np.int32)
        bc = np.zeros(max(self.nnzb(HSCHUR if self.schur else HPP), 1), np.int32)
        _check(self.L.g2ohip_get_partition(self.h, _ip(po), _ip(bc)), "getPartition")
        return po, bc[:self.nnzb(HSCHUR if self.schur else HPP)]

    def partitionPoses(self, colptr, rowidx, world):
        """Host-only pose partition of an upper block-CCS pattern, with this solver's ordering options."""
        return partition_poses(self.p, colptr, rowidx, world, self.h)

    def solveReducedLocal(self):
        _check(self.L.g2ohip_solve_reduced_local(self.h), "solveReducedLocal")

    def solveReducedShared(self):
        _check(self.L.g2ohip_solve_reduced_shared(self.h), "solveReducedShared")

    def solveReducedFinish(self):
        return _check(self.L.g2ohip_solve_reduced_finish(self.h), "solveReducedFinish") == OK

    def schurOperatorPrepare(self):
        _check(self.L.g2ohip_schur_operator_prepare(self.h), "schurOperatorPrepare")

    def schurOperatorApply(self, in_ptr, out_ptr):
        """out = (Hpp + lambda I - Hpl Dinv Hpl') in on raw device pointers (nP * p doubles each)."""
        _check(self.L.g2ohip_schur_operator_apply(self.h, C.c_void_p(int(in_ptr)), C.c_void_p(int(out_ptr))), "schurOperatorApply")

    def solveAsync(self):
        _check(self.L.g2ohip_solve_async(self.h), "solveAsync")

    def trialStatsBegin(self, lam):
        """Queue the sums of trialStats and their read-back without waiting; the next trialStats returns them."""
        _check(self.L.g2ohip_trial_stats_begin(self.h, float(lam)), "trialStatsBegin")

    def trialStats(self, lam):
        """(solve status of a pending solveAsync, chi2, computeScale(lam)) behind one synchronisation."""
        ok, chi, sc = C.c_int(1), C.c_double(0.0), C.c_double(0.0)
        _check(self.L.g2ohip_trial_stats(self.h, float(lam), C.byref(ok), C.byref(chi), C.byref(sc)), "trialStats")
        if ok.value == 2:
            return None, chi.value, sc.value      # repeat the trial (see g2ohip_trial_stats)
        return bool(ok.value), chi.value, sc.value

    def solveReducedFinishAsync(self):
        _check(self.L.g2ohip_solve_reduced_finish_async(self.h), "solveReducedFinishAsync")

    def exchangeSetup(self, block_idx, block_keep, pose_idx, pose_keep, halo_idx, halo_mine):
        bi, pi, hi = _i32(block_idx), _i32(pose_idx), _i32(halo_idx)
        bk, pk, hm = _f64(block_keep), _f64(pose_keep), _f64(halo_mine)
        _check(self.L.g2ohip_exchange_setup(self.h, len(bi), _ip(bi), _dp(bk), len(pi), _ip(pi), _dp(pk), len(hi), _ip(hi), _dp(hm)),
               "exchangeSetup")

    def exchangePack(self, which):
        _check(self.L.g2ohip_exchange_pack(self.h, which), "exchangePack")

    def exchangeUnpack(self, which):
        _check(self.L.g2ohip_exchange_unpack(self.h, which), "exchangeUnpack")

    def exchangeStatus(self):
        """True: solved; False: not positive definite on some rank; REPEAT (2): run the phases of this solve again on every
        rank (a dependency-driven launch gave up waiting somewhere; include/g2ohip.h G2OHIP_REPEAT)."""
        rc = _check(self.L.g2ohip_exchange_status(self.h), "exchangeStatus")
        return REPEAT if rc == REPEAT else rc == OK

    def solveBackSubstitute(self):
        _check(self.L.g2ohip_solve_back_substitute(self.h), "solveBackSubstitute")

    # ---- collectives inside the library (g2ohip_comm_*) ----
    @staticmethod
    def commUniqueId():
        """128 opaque bytes (ncclGetUniqueId); rank 0 creates them, every rank passes them to commInitRccl."""
        buf = C.create_string_buffer(128)
        _check(load().g2ohip_comm_unique_id(buf), "commUniqueId")
        return buf.raw

    def commInitRccl(self, rank, world, unique_id):
        _check(self.L.g2ohip_comm_init_rccl(self.h, rank, world, C.create_string_buffer(bytes(unique_id), 128)), "commInitRccl")

    def commInitHost(self, rank, world, fn):
        """fn(numpy view of the host buffer, op) reduces IN PLACE over the ranks (op 0 = sum, 1 = max)."""
        def tramp(ctx, ptr, n, op):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(n,)), op)
                return 0
            except Exception:      # noqa: BLE001 -- a Python exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._host_comm_cb = HOST_ALLREDUCE_FN(tramp)      # keep the trampoline alive as long as the handle
        _check(self.L.g2ohip_comm_init_host(self.h, rank, world, self._host_comm_cb, None), "commInitHost")

    def commInitPeer(self, rank, world, fn, slot_doubles=1 << 18):
        """Peer mailboxes over hipIpc handles (opt-in, see g2ohip.h); fn as in commInitHost: it carries the handles once and the
        host scalars afterwards."""
        def tramp(ctx, ptr, n, op):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(n,)), op)
                return 0
            except Exception:      # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1
        self._host_comm_cb = HOST_ALLREDUCE_FN(tramp)
        _check(self.L.g2ohip_comm_init_peer(self.h, rank, world, self._host_comm_cb, None, int(slot_doubles)), "commInitPeer")

    def commDestroy(self):
        _check(self.L.g2ohip_comm_destroy(self.h), "commDestroy")

    def commAllReduce(self, device_ptr, count, op=0):
        _check(self.L.g2ohip_comm_all_reduce(self.h, C.c_void_p(int(device_ptr)), int(count), int(op)), "commAllReduce")

    def solveSharded(self):
        """The whole sharded linear solve with the collectives inside the library; False = not positive definite."""
        return _check(self.L.g2ohip_solve_sharded(self.h), "solveSharded") == OK

    def chi2Sharded(self):
        v = C.c_double()
        _check(self.L.g2ohip_chi2_sharded(self.h, C.byref(v)), "chi2Sharded")
        return v.value

    def maxDiagonalSharded(self):
        v = C.c_double()
        _check(self.L.g2ohip_max_diagonal_sharded(self.h, C.byref(v)), "maxDiagonalSharded")
        return v.value

    def computeScaleSharded(self, lam):
        v = C.c_double()
        _check(self.L.g2ohip_compute_scale_sharded(self.h, lam, C.byref(v)), "computeScaleSharded")
        return v.value

    def vectorSize(self):
        return self.L.g2ohip_vector_size(self.h)

    def x(self):
        out = np.empty(self.vectorSize())
        _check(self.L.g2ohip_copy_x(self.h, _dp(out)), "x")
        return out

    def b(self):
        out = np.empty(self.vectorSize())
        _check(self.L.g2ohip_copy_b(self.h, _dp(out)), "b")
        return out

    def setX(self, x):
        x = _f64(x)
        if len(x) != self.vectorSize():
            raise ValueError("setX: wrong vector length")
        _check(self.L.g2ohip_set_x(self.h, _dp(x)), "setX")

    def multiplyHessian(self, src):
        src = _f64(src)
        dst = np.zeros_like(src)
        _check(self.L.g2ohip_multiply_hessian(self.h, _dp(dst), _dp(src)), "multiplyHessian")
        return dst

    def sync(self):
        _check(self.L.g2ohip_sync(self.h), "sync")

    def setStream(self, raw_stream):
        _check(self.L.g2ohip_set_stream(self.h, C.c_void_p(raw_stream)), "setStream")

    def clearEdgeSets(self):
        _check(self.L.g2ohip_clear_edge_sets(self.h), "clearEdgeSets")
        self._set_sizes = {}
        self.nP = self.nL = 0

    def updateStructure(self, new_poses, set_id=-1, v0=None, v1=None):
        """Solver::updateStructure (block_solver.hpp:297-351): new pose vertices behind the existing ones, new edges appended to
        edge set set_id.  False where the reference refuses (Schur complement active)."""
        n = 0 if v0 is None else len(v0)
        a0 = _i32(v0) if n else None
        a1 = _i32(v1) if (n and v1 is not None) else None
        rc = self.L.g2ohip_update_structure(self.h, int(new_poses), int(set_id), n, a0.ctypes.data if n else None,
                                            a1.ctypes.data if a1 is not None else None)
        if rc == ERR_UNSUPPORTED:
            return False
        _check(rc, "updateStructure")
        self.nP += int(new_poses)
        if n:
            self._set_sizes[set_id] += n
        return True

    def setProfiling(self, on):
        self.L.g2ohip_set_profiling(self.h, int(on))

    def setOption(self, name, value):
        _check(self.L.g2ohip_set_option(self.h, name.encode(), float(value)), "setOption")

    def stats(self):
        s = Stats()
        _check(self.L.g2ohip_get_stats(self.h, C.byref(s)), "stats")
        return s.as_dict()

    # ---- device-resident bundle-adjustment front end (EdgeProjectXYZ2UV graphs) ----------------
    def baSetEdges(self, set_id, cam_vertex, point_vertex, meas, info=None, f=1000.0, cx=320.0, cy=240.0):
        cv, pv, m = _i32(cam_vertex), _i32(point_vertex), _f64(meas)
        inf = None if info is None else _f64(info)
        n = self._set_sizes[set_id]
        if len(cv) != n or len(pv) != n or m.size != 2 * n or (inf is not None and inf.size != 4 * n):
            raise ValueError("baSetEdges: arrays must hold one entry per edge of set %d (%d edges)" % (set_id, n))
        _check(self.L.g2ohip_ba_set_edges(self.h, set_id, _ip(cv), _ip(pv), _dp(m), None if inf is None else _dp(inf),
                                          f, cx, cy), "baSetEdges")

    def baSetEdgesClasses(self, set_id, cam_vertex, point_vertex, meas, class_params, edge_class, info=None):
        """class_params [n_classes][5] = (f, cx, cy, robust kernel kind, delta); edge_class [n] (g2ohip_ba_set_edges_classes)."""
        cv, pv, m = _i32(cam_vertex), _i32(point_vertex), _f64(meas)
        inf = None if info is None else _f64(info)
        cp, ec = _f64(class_params), _i32(edge_class)
        n = self._set_sizes[set_id]
        if len(cv) != n or len(pv) != n or m.size != 2 * n or len(ec) != n or cp.size % 5 or (inf is not None and inf.size != 4 * n):
            raise ValueError("baSetEdgesClasses: arrays must hold one entry per edge of set %d (%d edges)" % (set_id, n))
        _check(self.L.g2ohip_ba_set_edges_classes(self.h, set_id, _ip(cv), _ip(pv), _dp(m), None if inf is None else _dp(inf),
                                                  cp.size // 5, _dp(cp), _ip(ec)), "baSetEdgesClasses")

    def baSetEstimates(self, cams, cam_hidx, points, point_hidx):
        cams, points = _f64(cams), _f64(points)
        ch, ph = _i32(cam_hidx), _i32(point_hidx)
        self._ba_n = (len(ch), len(ph))
        _check(self.L.g2ohip_ba_set_estimates(self.h, len(ch), _dp(cams), _ip(ch), len(ph), _dp(points), _ip(ph)),
               "baSetEstimates")

    def baGetEstimates(self):
        cams = np.empty((self._ba_n[0], 12))
        pts = np.empty((self._ba_n[1], 3))
        _check(self.L.g2ohip_ba_get_estimates(self.h, _dp(cams), _dp(pts)), "baGetEstimates")
        return cams, pts

    def baGetEstimatesOf(self, cam_index, point_index):
        """Estimates of selected cameras / points (indices into the arrays of baSetEstimates)."""
        ci, pi = _i32(cam_index), _i32(point_index)
        cams, pts = np.empty((len(ci), 12)), np.empty((len(pi), 3))
        _check(self.L.g2ohip_ba_get_estimates_of(self.h, len(ci), _ip(ci), _dp(cams), len(pi), _ip(pi), _dp(pts)), "baGetEstimatesOf")
        return cams, pts

    def baFetchEstimatesBegin(self, pieces=4):
        """Asynchronous read-back of the estimates in 1 + pieces pieces (cameras, then point ranges); baFetchEstimatesWait(k)
        returns the arrays once piece k has arrived (the arrays are complete after the last piece)."""
        self._fetch = (np.empty((self._ba_n[0], 12)), np.empty((self._ba_n[1], 3)), pieces)
        _check(self.L.g2ohip_ba_fetch_estimates_begin(self.h, _dp(self._fetch[0]), _dp(self._fetch[1]), pieces), "baFetchEstimatesBegin")

    def baFetchEstimatesWait(self, piece):
        _check(self.L.g2ohip_ba_fetch_estimates_wait(self.h, piece), "baFetchEstimatesWait")
        return self._fetch[0], self._fetch[1]

    def baLinearize(self, jacobians=True):
        _check(self.L.g2ohip_ba_linearize(self.h, int(jacobians)), "baLinearize")

    def baUpdate(self):
        _check(self.L.g2ohip_ba_update(self.h), "baUpdate")

    def baPush(self):
        _check(self.L.g2ohip_ba_push(self.h), "baPush")

    def baPop(self):
        _check(self.L.g2ohip_ba_pop(self.h), "baPop")

    def baDiscardTop(self):
        _check(self.L.g2ohip_ba_discard_top(self.h), "baDiscardTop")

    def kernelTimes(self, reset=True):
        """{kernel name: (total seconds, launches)} from HIP events on the solver stream."""
        out = {}
        for k in range(self.L.g2ohip_kernel_slots()):
            t = C.c_double()
            n = C.c_long()
            _check(self.L.g2ohip_kernel_time(self.h, k, C.byref(t), C.byref(n), int(reset)), "kernelTimes")
            if n.value:
                out[self.L.g2ohip_kernel_name(k).decode()] = (t.value, n.value)
        return out

    # ---- inspection --------------------------------------------------------------------
    def nnzb(self, which):
        v = C.c_int()
        _check(self.L.g2ohip_get_nnzb(self.h, which, C.byref(v)), "nnzb")
        return v.value

    def pattern(self, which):
        ncols = self.nL if which == HPL else self.nP
        cp = np.zeros(ncols + 1, np.int32)
        ri = np.zeros(max(self.nnzb(which), 1), np.int32)
        _check(self.L.g2ohip_get_pattern(self.h, which, _ip(cp), _ip(ri)), "pattern")
        return cp, ri[:self.nnzb(which)]

    def values(self, which):
        p, l = self.p, self.l
        per = {HPP: p * p, HPL: p * l, HLL: l * l, HSCHUR: p * p, DINV: l * l}[which]
        out = np.empty(max(self.nnzb(which) * per, 1))
        _check(self.L.g2ohip_copy_values(self.h, which, _dp(out)), "values")
        return out[:self.nnzb(which) * per]

    def computeMarginals(self, rows, cols):
        """Blocks (rows[i], cols[i]) of the inverse of the (reduced) pose system: [n][p][p] (row, column) or None if not PD."""
        rows, cols = _i32(rows), _i32(cols)
        out = np.empty((len(rows), self.p * self.p))
        rc = _check(self.L.g2ohip_compute_marginals(self.h, len(rows), _ip(rows), _ip(cols), _dp(out)), "computeMarginals")
        if rc != OK:
            return None
        return out.reshape(len(rows), self.p, self.p).transpose(0, 2, 1).copy()   # column-major blocks -> [row][col]

    def diagonal(self):
        """Scalar diagonal of H with the current damping (poses then landmarks): g2ohip_copy_diagonal."""
        out = np.empty(self.vectorSize())
        _check(self.L.g2ohip_copy_diagonal(self.h, _dp(out)), "diagonal")
        return out

    def edgeData(self, set_id, n, d, d0, d1):
        """(J0, J1, err) of a binary edge set (n edges, error dim d, vertex dims d0 / d1) as the next buildSystem reads them."""
        J0, J1, err = np.empty((n, d * d0)), np.empty((n, d * d1)), np.empty((n, d))
        _check(self.L.g2ohip_copy_edge_data(self.h, set_id, _dp(J0), _dp(J1), _dp(err)), "edgeData")
        return J0, J1, err

    # ---- device-resident pose-graph front end (EdgeSE2 = 1, EdgeSE3 = 2) ------------------------------
    def pgSetEdges(self, set_id, edge_type, vi, vj, meas, info):
        vi, vj, meas, info = _i32(vi), _i32(vj), _f64(meas), _f64(info)
        self._pg = (edge_type, 3 if edge_type == 1 else 12)
        n, d = self._set_sizes[set_id], (3 if edge_type == 1 else 6)
        if len(vi) != n or len(vj) != n or meas.size != n * self._pg[1] or info.size != n * d * d:
            raise ValueError("pgSetEdges: arrays must hold one entry per edge of set %d (%d edges)" % (set_id, n))
        _check(self.L.g2ohip_pg_set_edges(self.h, set_id, edge_type, _ip(vi), _ip(vj), _dp(meas), _dp(info)), "pgSetEdges")

    def pgSetEstimates(self, poses, hidx):
        poses, hidx = _f64(poses), _i32(hidx)
        self._pg_nv = len(hidx)
        _check(self.L.g2ohip_pg_set_estimates(self.h, len(hidx), _dp(poses), _ip(hidx)), "pgSetEstimates")

    def pgGetEstimates(self):
        out = np.empty((self._pg_nv, self._pg[1]))
        _check(self.L.g2ohip_pg_get_estimates(self.h, _dp(out)), "pgGetEstimates")
        return out

    def pgLinearize(self, jacobians=True):
        _check(self.L.g2ohip_pg_linearize(self.h, int(jacobians)), "pgLinearize")

    def pgUpdate(self):
        _check(self.L.g2ohip_pg_update(self.h), "pgUpdate")

    def pgPush(self):
        _check(self.L.g2ohip_pg_push(self.h), "pgPush")

    def pgPop(self):
        _check(self.L.g2ohip_pg_pop(self.h), "pgPop")

    def pgDiscardTop(self):
        _check(self.L.g2ohip_pg_discard_top(self.h), "pgDiscardTop")

    def deviceArray(self, which):
        ptr = C.c_void_p()
        n = C.c_size_t()
        _check(self.L.g2ohip_device_array(self.h, which, C.byref(ptr), C.byref(n)), "deviceArray")
        return ptr.value, n.value


def partition_poses(block_dim, colptr, rowidx, world, options_from=None):
    """g2ohip_partition_poses: (owner rank of every block column, consumer rank of every block of the
    pattern); -1 = shared top of the elimination tree.  Host only, no device."""
    L = load()
    colptr, rowidx = _i32(colptr), _i32(rowidx)
    nb = len(colptr) - 1
    po = np.zeros(nb, np.int32)
    bc = np.zeros(max(len(rowidx), 1), np.int32)
    _check(L.g2ohip_partition_poses(options_from, block_dim, nb, _ip(colptr), _ip(rowidx), world, _ip(po), _ip(bc)),
           "partitionPoses")
    return po, bc[:len(rowidx)]


class HipLinearSolver:
    """g2o::LinearSolver<MatrixType> (linear_solver.h:40-81) over the multifrontal engine."""

    def __init__(self, block_dim, device=0):
        self.L = load()
        self.bs = block_dim
        self.h = C.c_void_p()
        _check(self.L.g2ohip_ls_create(C.byref(self.h), block_dim, device), "ls_create")

    def init(self):
        _check(self.L.g2ohip_ls_init(self.h), "ls_init")
        return True

    def setOption(self, name, value):
        _check(self.L.g2ohip_ls_set_option(self.h, name.encode(), float(value)), "ls_set_option")

    def solve(self, colptr, rowidx, values, b):
        """Returns (ok, x).  A = upper block CCS."""
        colptr, rowidx, values, b = _i32(colptr), _i32(rowidx), _f64(values), _f64(b)
        nb = len(colptr) - 1
        x = np.zeros(nb * self.bs)
        rc = _check(self.L.g2ohip_ls_solve(self.h, nb, _ip(colptr), _ip(rowidx), _dp(values), _dp(x), _dp(b)), "ls_solve")
        return rc == OK, x

    def solvePattern(self, colptr, rowidx, values, rows, cols):
        """LinearSolver::solvePattern: blocks (rows[i], cols[i]) of A^-1 as [n][bd][bd] (row, column); None if not PD."""
        colptr, rowidx, rows, cols = _i32(colptr), _i32(rowidx), _i32(rows), _i32(cols)
        values = _f64(values)
        nb = len(colptr) - 1
        bd = self.bs
        if len(rows) != len(cols) or len(rowidx) < colptr[nb] or values.size < colptr[nb] * bd * bd:
            raise ValueError("solvePattern: inconsistent array lengths")
        out = np.empty((len(rows), bd * bd))
        rc = _check(self.L.g2ohip_ls_solve_pattern(self.h, nb, _ip(colptr), _ip(rowidx), _dp(values), len(rows), _ip(rows), _ip(cols), _dp(out)),
                    "ls_solve_pattern")
        if rc != OK:
            return None
        return out.reshape(len(rows), bd, bd).transpose(0, 2, 1).copy()

    def stats(self):
        s = Stats()
        _check(self.L.g2ohip_ls_get_stats(self.h, C.byref(s)), "ls_stats")
        return s.as_dict()

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.g2ohip_ls_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
