"""TEST INFRASTRUCTURE ONLY -- ctypes loader for the CPU oracle (oracle/libg2o_oracle.so)
and, when present, the reference's own compiled CSparse path (oracle/_ref/).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (openslam_g2o_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int_p)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dbl_p)


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(HERE, "libg2o_oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(
            os.path.getmtime(os.path.join(HERE, f)) for f in ("g2o_oracle.c", "g2o_oracle_types.c")):
        subprocess.check_call(["make", "-C", HERE, "-s", os.path.join(HERE, "libg2o_oracle.so")])
    if os.path.isdir("/root/reference/EXTERNAL/csparse") and (force or not os.path.exists(
            os.path.join(HERE, "_ref", "libg2o_ref_csparse.so"))):
        subprocess.check_call(["make", "-C", HERE, "-s", "ref"])


_LIB_OMP = None


def lib(omp=False):
    """The oracle library; omp=True: the build with the reference's optional OpenMP regions enabled
    (libg2o_oracle_omp.so, "best CPU" timing only -- its summation order depends on the thread schedule)."""
    global _LIB, _LIB_OMP
    if omp:
        if _LIB_OMP is None:
            so = os.path.join(HERE, "libg2o_oracle_omp.so")
            src = [os.path.join(HERE, f) for f in ("g2o_oracle.c", "g2o_oracle_types.c")]
            if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in src):
                subprocess.check_call(["make", "-C", HERE, "-s", so])
            _LIB_OMP = _proto(C.CDLL(so))
        return _LIB_OMP
    if _LIB is None:
        build()
        _LIB = _proto(C.CDLL(os.path.join(HERE, "libg2o_oracle.so")))
    return _LIB


def _proto(L):
    if True:   # (prototypes)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int] * 5
        L.orc_add_edge_set.argtypes = [C.c_void_p, C.c_int, C.c_int, c_int_p, c_int_p]
        L.orc_set_dims.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_build_structure.argtypes = [C.c_void_p]
        L.orc_set_edge_data.argtypes = [C.c_void_p, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, C.c_double]
        L.orc_build_system.argtypes = [C.c_void_p]
        L.orc_add_multi_edge_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_int_p]
        L.orc_set_multi_edge_data.argtypes = [C.c_void_p, C.c_int, C.POINTER(c_dbl_p), c_dbl_p, c_dbl_p, C.c_double, C.c_int]
        L.orc_chi2.restype = C.c_double
        L.orc_chi2.argtypes = [C.c_void_p]
        L.orc_set_robust_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_robustify.argtypes = [C.c_int, C.c_double, C.c_double, c_dbl_p]
        L.orc_set_lambda.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.orc_restore_diagonal.argtypes = [C.c_void_p]
        L.orc_set_lambda_split.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        L.orc_add_schur_pattern.argtypes = [C.c_void_p, C.c_int, c_int_p, c_int_p]
        L.orc_solve_schur.argtypes = [C.c_void_p]
        L.orc_solve_reduced.argtypes = [C.c_void_p]
        L.orc_solve_back_substitute.argtypes = [C.c_void_p]
        L.orc_max_diagonal.restype = C.c_double
        L.orc_max_diagonal.argtypes = [C.c_void_p]
        L.orc_compute_scale.restype = C.c_double
        L.orc_compute_scale.argtypes = [C.c_void_p, C.c_double]
        L.orc_set_ordering.argtypes = [C.c_void_p, C.c_int, c_int_p]
        L.orc_solve.argtypes = [C.c_void_p]
        L.orc_multiply_full.argtypes = [C.c_void_p, c_dbl_p, c_dbl_p]
        L.orc_destroy.argtypes = [C.c_void_p]
        for nm in ("x", "b", "bschur", "Hpp", "Hpl", "Hll", "Hschur", "Dinv"):
            f = getattr(L, "orc_" + nm)
            f.restype = c_dbl_p
            f.argtypes = [C.c_void_p]
        for nm in ("pp_colptr", "pp_row", "pl_colptr", "pl_row", "hs_colptr", "hs_row"):
            f = getattr(L, "orc_" + nm)
            f.restype = c_int_p
            f.argtypes = [C.c_void_p]
        for nm in ("pp_nnzb", "pl_nnzb", "hs_nnzb"):
            getattr(L, "orc_" + nm).argtypes = [C.c_void_p]
        L.orc_lnz.restype = C.c_double
        L.orc_lnz.argtypes = [C.c_void_p]
        L.orc_time.restype = C.c_double
        L.orc_time.argtypes = [C.c_void_p, C.c_int]
        L.orc_linear_solve_blocks.argtypes = [C.c_int, C.c_int, c_int_p, c_int_p, c_dbl_p, c_int_p, C.c_int,
                                              c_dbl_p, c_dbl_p, c_dbl_p]
        L.orc_fill_scalar_ccs.restype = C.c_long
        L.orc_fill_scalar_ccs.argtypes = [C.c_int, C.c_int, c_int_p, c_int_p, c_dbl_p, c_int_p, c_int_p, c_dbl_p]
        L.orc_se2_edges.argtypes = [C.c_int, c_dbl_p, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p]
        L.orc_se2_oplus.argtypes = [C.c_int, c_dbl_p, c_int_p, c_dbl_p]
        L.orc_ba_edges.argtypes = [C.c_int, c_dbl_p, c_dbl_p, c_int_p, c_int_p, c_dbl_p, C.c_double, C.c_double,
                                   C.c_double, c_dbl_p, c_dbl_p, c_dbl_p]
        L.orc_ba_oplus_cams.argtypes = [C.c_int, c_dbl_p, c_int_p, c_dbl_p]
        L.orc_ba_oplus_pts.argtypes = [C.c_int, c_dbl_p, c_int_p, c_dbl_p]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
    return L


def ref():
    """The reference's own CSparse Cholesky path (None when oracle/_ref was never built)."""
    global _REF
    if _REF is None:
        so = os.path.join(HERE, "_ref", "libg2o_ref_csparse.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        R.ref_block_amd.argtypes = [C.c_int, c_int_p, c_int_p, c_int_p]
        R.ref_scalar_amd.argtypes = [C.c_int, c_int_p, c_int_p, c_int_p]
        R.ref_symbolic.restype = C.c_void_p
        R.ref_symbolic.argtypes = [C.c_int, c_int_p, c_int_p, c_int_p]
        R.ref_lnz.restype = C.c_double
        R.ref_lnz.argtypes = [C.c_void_p]
        R.ref_cholsolve.argtypes = [C.c_void_p, c_int_p, c_int_p, c_dbl_p, c_dbl_p]
        R.ref_free.argtypes = [C.c_void_p]
        _REF = R
    return _REF


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleSolver:
    """Mirror of g2o::BlockSolver<p,l> (g2o/core/block_solver.h:98-178) over flat arrays."""

    def __init__(self, p, l, nP, nL, schur=True, omp=False):
        self.L = lib(omp)
        self.p, self.l, self.nP, self.nL = p, l, nP, nL
        self.h = C.c_void_p(self.L.orc_create(p, l, nP, nL, int(schur)))
        self.schur = bool(schur) and nL > 0
        self._keep = []
        self.sets = []

    def add_edge_set(self, d, v0, v1=None):
        v0 = _i32(v0)
        v1 = None if v1 is None else _i32(v1)
        s = self.L.orc_add_edge_set(self.h, d, len(v0), _ip(v0), _ip(v1))
        assert s >= 0

        def cls_dim(v):
            free = v[v >= 0]
            if len(free) == 0:
                return None
            return self.p if free[0] < self.nP else self.l
        self.sets.append(dict(d=d, n=len(v0), v0=v0, v1=v1))
        return s

    def set_dims(self, s, dim0, dim1):
        self.sets[s]["dim0"], self.sets[s]["dim1"] = dim0, dim1
        self.L.orc_set_dims(self.h, s, dim0, dim1)

    def build_structure(self):
        rc = self.L.orc_build_structure(self.h)
        assert rc == 0, rc

    def set_edge_data(self, s, J0, J1, omega, err, huber_delta=0.0):
        J0, omega, err = _f64(J0), _f64(omega), _f64(err)
        J1 = None if J1 is None else _f64(J1)
        self._keep = [k for k in self._keep if k[0] != s] + [(s, J0, J1, omega, err)]
        self.L.orc_set_edge_data(self.h, s, _dp(J0), _dp(J1), _dp(omega), _dp(err), float(huber_delta))

    def add_multi_edge_set(self, d, verts):
        """n-ary edges (BaseMultiEdge): verts [n][arity] hessian indices (-1 fixed).  Before build_structure."""
        verts = _i32(verts)
        m = self.L.orc_add_multi_edge_set(self.h, d, verts.shape[0], verts.shape[1], _ip(verts))
        assert m >= 0
        return m

    def set_multi_edge_data(self, m, J, omega, err, delta=0.0, kind=1):
        """J: one [n][d * dim_i] array (column-major d x dim_i Jacobians) per vertex position."""
        J = [_f64(j) for j in J]
        omega, err = _f64(omega), _f64(err)
        ptrs = (c_dbl_p * len(J))(*[_dp(j) for j in J])
        self._keep.append(("m%d" % m, J, omega, err, ptrs))
        self.L.orc_set_multi_edge_data(self.h, m, ptrs, _dp(omega), _dp(err), float(delta), int(kind))

    def set_robust_kernel(self, s, kind):
        """1 Huber (default), 2 PseudoHuber, 3 Cauchy, 4 Saturated, 5 DCS; delta = the huber_delta of set_edge_data."""
        self.L.orc_set_robust_kernel(self.h, s, int(kind))

    def build_system(self):
        self.L.orc_build_system(self.h)

    def chi2(self):
        return self.L.orc_chi2(self.h)

    def set_lambda(self, lam, backup=False):
        self.L.orc_set_lambda(self.h, lam, int(backup))

    def restore_diagonal(self):
        self.L.orc_restore_diagonal(self.h)

    def set_lambda_split(self, lam_pose, lam_landmark, backup=False):
        self.L.orc_set_lambda_split(self.h, lam_pose, lam_landmark, int(backup))

    def add_schur_pattern(self, rows, cols):
        rows, cols = _i32(rows), _i32(cols)
        self.L.orc_add_schur_pattern(self.h, len(rows), _ip(rows), _ip(cols))

    def solve_schur(self):
        self.L.orc_solve_schur(self.h)

    def solve_reduced(self):
        return bool(self.L.orc_solve_reduced(self.h))

    def solve_back_substitute(self):
        self.L.orc_solve_back_substitute(self.h)

    def view(self, name, n):
        """Zero-copy numpy view of an oracle-owned array (for in-place reductions in tests)."""
        ptr = getattr(self.L, "orc_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n,))

    def max_diagonal(self):
        return self.L.orc_max_diagonal(self.h)

    def compute_scale(self, lam):
        return self.L.orc_compute_scale(self.h, lam)

    def set_ordering(self, mode, block_perm=None):
        bp = None if block_perm is None else _i32(block_perm)
        self.L.orc_set_ordering(self.h, mode, _ip(bp))

    def solve(self):
        return bool(self.L.orc_solve(self.h))

    def _arr(self, name, n):
        ptr = getattr(self.L, "orc_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n,)) if n > 0 else np.zeros(0)

    @property
    def n(self):
        return self.p * self.nP + self.l * self.nL

    def x(self):
        return self._arr("x", self.n).copy()

    def b(self):
        return self._arr("b", self.n).copy()

    def bschur(self):
        return self._arr("bschur", self.p * self.nP).copy()

    def pattern(self, which):
        ncols = self.nL if which == "pl" else self.nP
        cp = np.ctypeslib.as_array(getattr(self.L, "orc_%s_colptr" % which)(self.h), shape=(ncols + 1,)).copy()
        nnzb = getattr(self.L, "orc_%s_nnzb" % which)(self.h)
        row = np.ctypeslib.as_array(getattr(self.L, "orc_%s_row" % which)(self.h), shape=(max(nnzb, 1),)).copy()[:nnzb]
        return cp, row

    def values(self, which):
        p, l = self.p, self.l
        if which == "Hpp":
            return self._arr("Hpp", self.L.orc_pp_nnzb(self.h) * p * p).copy()
        if which == "Hpl":
            return self._arr("Hpl", self.L.orc_pl_nnzb(self.h) * p * l).copy()
        if which == "Hll":
            return self._arr("Hll", self.nL * l * l).copy()
        if which == "Hschur":
            return self._arr("Hschur", self.L.orc_hs_nnzb(self.h) * p * p).copy()
        if which == "Dinv":
            return self._arr("Dinv", self.nL * l * l).copy()
        raise KeyError(which)

    def multiply_full(self, src):
        src = _f64(src)
        dst = np.zeros_like(src)
        self.L.orc_multiply_full(self.h, _dp(src), _dp(dst))
        return dst

    def lnz(self):
        return self.L.orc_lnz(self.h)

    def times(self):
        return dict(schur=self.L.orc_time(self.h, 0), linear=self.L.orc_time(self.h, 1),
                    numeric=self.L.orc_time(self.h, 2))

    def dense_full(self):
        """Dense symmetric [Hpp Hpl; Hpl' Hll] for small cross-checks."""
        p, l, nP, nL = self.p, self.l, self.nP, self.nL
        n = self.n
        H = np.zeros((n, n))
        cp, row = self.pattern("pp")
        v = self.values("Hpp").reshape(-1, p, p)
        for c in range(nP):
            for q in range(cp[c], cp[c + 1]):
                r = row[q]
                B = v[q].T  # stored column-major
                H[r * p:(r + 1) * p, c * p:(c + 1) * p] = B
                H[c * p:(c + 1) * p, r * p:(r + 1) * p] = B.T
        if nL:
            cp, row = self.pattern("pl")
            v = self.values("Hpl").reshape(-1, l, p)
            o = p * nP
            for c in range(nL):
                for q in range(cp[c], cp[c + 1]):
                    r = row[q]
                    B = v[q].T
                    H[r * p:(r + 1) * p, o + c * l:o + (c + 1) * l] = B
                    H[o + c * l:o + (c + 1) * l, r * p:(r + 1) * p] = B.T
            v = self.values("Hll").reshape(-1, l, l)
            for c in range(nL):
                H[o + c * l:o + (c + 1) * l, o + c * l:o + (c + 1) * l] = v[c].T
        return H

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass


def linear_solve_blocks(nb, bs, colptr, row, val, b, block_perm=None, min_degree=True):
    L = lib()
    colptr, row, val, b = _i32(colptr), _i32(row), _f64(val), _f64(b)
    x = np.zeros(nb * bs)
    lnz = C.c_double(0)
    bp = None if block_perm is None else _i32(block_perm)
    ok = L.orc_linear_solve_blocks(nb, bs, _ip(colptr), _ip(row), _dp(val), _ip(bp), int(min_degree), _dp(b), _dp(x),
                                   C.byref(lnz))
    return bool(ok), x, lnz.value


def scalar_ccs(nb, bs, colptr, row, val):
    L = lib()
    colptr, row, val = _i32(colptr), _i32(row), _f64(val)
    nz = L.orc_fill_scalar_ccs(nb, bs, _ip(colptr), _ip(row), _dp(val), None, None, None)
    Ap = np.zeros(nb * bs + 1, np.int32)
    Ai = np.zeros(nz, np.int32)
    Ax = np.zeros(nz)
    L.orc_fill_scalar_ccs(nb, bs, _ip(colptr), _ip(row), _dp(val), _ip(Ap), _ip(Ai), _dp(Ax))
    return Ap, Ai, Ax


def ref_solve_blocks(nb, bs, colptr, row, val, b, block_ordering=True):
    """Reference path: fillCCS (restated) -> cs_amd on the block pattern -> scalar blow-up
    -> reference symbolic + csparse_extension::cs_cholsolsymb.  Returns (ok, x, lnz, block_perm)."""
    R = ref()
    assert R is not None, "oracle/_ref not built"
    Ap, Ai, Ax = scalar_ccs(nb, bs, colptr, row, val)
    colptr, row = _i32(colptr), _i32(row)
    n = nb * bs
    if block_ordering:
        P = np.zeros(nb, np.int32)
        assert R.ref_block_amd(nb, _ip(colptr), _ip(row), _ip(P))
        sperm = (P[:, None] * bs + np.arange(bs, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)
    else:
        P = None
        sperm = np.zeros(n, np.int32)
        assert R.ref_scalar_amd(n, _ip(Ap), _ip(Ai), _ip(sperm))
    h = C.c_void_p(R.ref_symbolic(n, _ip(Ap), _ip(Ai), _ip(sperm)))
    x = _f64(b).copy()
    ok = R.ref_cholsolve(h, _ip(Ap), _ip(Ai), _dp(Ax), _dp(x))
    lnz = R.ref_lnz(h)
    R.ref_free(h)
    return bool(ok), x, lnz, P


# ---- input producers (restated types) ---------------------------------------------------
def se2_edges(poses, vi, vj, meas, jac=True):
    L = lib()
    poses, vi, vj, meas = _f64(poses), _i32(vi), _i32(vj), _f64(meas)
    n = len(vi)
    err = np.zeros((n, 3))
    if jac:
        J0 = np.zeros((n, 9))
        J1 = np.zeros((n, 9))
        L.orc_se2_edges(n, _dp(poses), _ip(vi), _ip(vj), _dp(meas), _dp(J0), _dp(J1), _dp(err))
        return J0, J1, err
    L.orc_se2_edges(n, _dp(poses), _ip(vi), _ip(vj), _dp(meas), None, None, _dp(err))
    return err


def se2_oplus(poses, hidx, x):
    L = lib()
    poses = _f64(poses).copy()
    hidx, x = _i32(hidx), _f64(x)
    L.orc_se2_oplus(len(hidx), _dp(poses), _ip(hidx), _dp(x))
    return poses


def ba_edges(cams, pts, cam_idx, pt_idx, meas, f, cx, cy, jac=True, omp=False):
    L = lib(omp)
    cams, pts, cam_idx, pt_idx, meas = _f64(cams), _f64(pts), _i32(cam_idx), _i32(pt_idx), _f64(meas)
    n = len(cam_idx)
    err = np.zeros((n, 2))
    if jac:
        Jp = np.zeros((n, 6))
        Jc = np.zeros((n, 12))
        L.orc_ba_edges(n, _dp(cams), _dp(pts), _ip(cam_idx), _ip(pt_idx), _dp(meas), f, cx, cy, _dp(Jp), _dp(Jc), _dp(err))
        return Jp, Jc, err
    L.orc_ba_edges(n, _dp(cams), _dp(pts), _ip(cam_idx), _ip(pt_idx), _dp(meas), f, cx, cy, None, None, _dp(err))
    return err


def ba_oplus(cams, pts, cam_hidx, pt_hidx_local, x, sizeP):
    L = lib()
    cams, pts = _f64(cams).copy(), _f64(pts).copy()
    x = _f64(x)
    cam_hidx, pt_hidx_local = _i32(cam_hidx), _i32(pt_hidx_local)
    xp, xl = np.ascontiguousarray(x[:sizeP]), np.ascontiguousarray(x[sizeP:])
    L.orc_ba_oplus_cams(len(cam_hidx), _dp(cams), _ip(cam_hidx), _dp(xp))
    L.orc_ba_oplus_pts(len(pt_hidx_local), _dp(pts), _ip(pt_hidx_local), _dp(xl))
    return cams, pts


# ---- 3-D pose graphs (VertexSE3 / EdgeSE3) ------------------------------------------------
def _se3_sigs():
    L = lib()
    if not getattr(L, "_se3_ready", False):
        L.orc_se3_from_qt.argtypes = [C.c_int, c_dbl_p, C.c_int, c_dbl_p]
        L.orc_se3_edges.argtypes = [C.c_int, c_dbl_p, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p]
        L.orc_se3_oplus.argtypes = [C.c_int, c_dbl_p, c_int_p, c_dbl_p]
        L._se3_ready = True
    return L


def se3_from_qt(qt, normalize=False):
    """[n][7] (x y z qx qy qz qw) -> [n][12] isometries (R column-major | t)."""
    L = _se3_sigs()
    qt = _f64(qt)
    T = np.zeros((len(qt), 12))
    L.orc_se3_from_qt(len(qt), _dp(qt), int(normalize), _dp(T))
    return T


def se3_edges(poses, vi, vj, meas, jac=True):
    L = _se3_sigs()
    poses, vi, vj, meas = _f64(poses), _i32(vi), _i32(vj), _f64(meas)
    n = len(vi)
    err = np.zeros((n, 6))
    if jac:
        J0 = np.zeros((n, 36))
        J1 = np.zeros((n, 36))
        L.orc_se3_edges(n, _dp(poses), _ip(vi), _ip(vj), _dp(meas), _dp(J0), _dp(J1), _dp(err))
        return J0, J1, err
    L.orc_se3_edges(n, _dp(poses), _ip(vi), _ip(vj), _dp(meas), None, None, _dp(err))
    return err


def se3_oplus(poses, hidx, x):
    L = _se3_sigs()
    poses = _f64(poses).copy()
    hidx, x = _i32(hidx), _f64(x)
    L.orc_se3_oplus(len(hidx), _dp(poses), _ip(hidx), _dp(x))
    return poses


def pcg_solve_blocks(nb, bs, colptr, row, val, b, tolerance=1e-6, absolute=True, max_iter=-1, residual=-1.0):
    """LinearSolverPCG::solve restated (oracle/g2o_oracle.c): returns (ok, x, iterations, residual_out)."""
    L = lib()
    colptr, row, val, b = _i32(colptr), _i32(row), _f64(val), _f64(b)
    x = np.zeros(nb * bs)
    res = C.c_double(residual)
    it = C.c_int(0)
    L.orc_pcg_solve_blocks.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_int, C.c_int,
                                       C.POINTER(C.c_double), C.POINTER(C.c_int)]
    ok = L.orc_pcg_solve_blocks(nb, bs, _ip(colptr), _ip(row), _dp(val), _dp(b), _dp(x), tolerance, int(absolute), max_iter,
                                C.byref(res), C.byref(it))
    return bool(ok), x, it.value, res.value


def robustify(kind, delta, e):
    """(rho, rho', rho'') of the reference's robust kernel `kind` at squared error e (robust_kernel_impl.cpp)."""
    L = lib()
    out = np.zeros(3)
    L.orc_robustify(int(kind), float(delta), float(e), out.ctypes.data_as(c_dbl_p))
    return out
