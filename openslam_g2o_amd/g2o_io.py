"""Minimal `.g2o` text reader for the pose-graph tags BASELINE.json's configs 1-2 use.

Mirrors what OptimizableGraph::load (g2o/core/optimizable_graph.cpp:356-480) does for
`VERTEX_SE2` / `EDGE_SE2` (g2o/types/slam2d/{vertex_se2,edge_se2}.cpp read()) and
`VERTEX_SE3:QUAT` / `EDGE_SE3:QUAT` (g2o/types/slam3d/{vertex_se3,edge_se3}.cpp read()):
the information matrix is given as its upper triangle, row-major (edge_se2.cpp:46-51).
Host-side bookkeeping only; nothing here is on the accelerated path.
"""
import numpy as np


def _upper_to_full(vals, d):
    M = np.zeros((d, d))
    k = 0
    for i in range(d):
        for j in range(i, d):
            M[i, j] = M[j, i] = vals[k]
            k += 1
    return M


def read_g2o(path):
    """Returns dict(kind='se2'|'se3', ids, estimates, edges=(vi, vj), meas, info, fixed)."""
    vid, vest, ei, ej, meas, info, fixed = [], [], [], [], [], [], []
    kind = None
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            tag = t[0]
            if tag == "VERTEX_SE2":
                kind = kind or "se2"
                vid.append(int(t[1]))
                vest.append([float(x) for x in t[2:5]])
            elif tag == "EDGE_SE2":
                ei.append(int(t[1]))
                ej.append(int(t[2]))
                meas.append([float(x) for x in t[3:6]])
                info.append(_upper_to_full([float(x) for x in t[6:12]], 3))
            elif tag == "VERTEX_SE3:QUAT":
                kind = kind or "se3"
                vid.append(int(t[1]))
                vest.append([float(x) for x in t[2:9]])
            elif tag == "EDGE_SE3:QUAT":
                ei.append(int(t[1]))
                ej.append(int(t[2]))
                meas.append([float(x) for x in t[3:10]])
                info.append(_upper_to_full([float(x) for x in t[10:31]], 6))
            elif tag == "FIX":
                fixed.extend(int(x) for x in t[1:])
    vid = np.asarray(vid, np.int64)
    order = np.argsort(vid, kind="stable")          # index mapping is by vertex id (sparse_optimizer.cpp:174-187)
    vid = vid[order]
    vest = np.asarray(vest, np.float64)[order]
    lut = {int(v): k for k, v in enumerate(vid)}
    vi = np.asarray([lut[a] for a in ei], np.int32)
    vj = np.asarray([lut[a] for a in ej], np.int32)
    return dict(kind=kind, ids=vid, estimates=vest, vi=vi, vj=vj, meas=np.asarray(meas, np.float64),
                info=np.asarray(info, np.float64), fixed=[lut[f] for f in fixed if f in lut])


def hessian_index(n_vertices, fixed):
    """buildIndexMapping (sparse_optimizer.cpp:166-190) for an all-pose graph: fixed -> -1."""
    h = np.full(n_vertices, -1, np.int32)
    k = 0
    fx = set(fixed)
    for v in range(n_vertices):
        if v not in fx:
            h[v] = k
            k += 1
    return h, k


# ---- bundle-adjustment tags --------------------------------------------------------------
# PARAMS_CAMERAPARAMETERS id f cx cy baseline          types_six_dof_expmap.h:62-76
# VERTEX_SE3:EXPMAP id tx ty tz qx qy qz qw            cam -> world; the vertex holds the inverse
#                                                      (types_six_dof_expmap.cpp:88-103, se3quat.h:138-153)
# VERTEX_XYZ id x y z                                  types_sba.cpp:40 (VertexSBAPointXYZ)
# EDGE_PROJECT_XYZ2UV:EXPMAP point pose param u v i00 i01 i11   (types_six_dof_expmap.cpp:241-270; the point is
#                                                      vertex 0, types_six_dof_expmap.h:133)
# FIX id ...                                           optimizable_graph.cpp:407-420
def _quat_to_R(q):
    """(x, y, z, w) -> rotation matrices [n][3][3] (Eigen's toRotationMatrix, no normalisation)."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _R_to_quat(R):
    """Rotation matrices -> unit quaternions (x, y, z, w), w >= 0 (Eigen's four-case conversion)."""
    q = np.empty((len(R), 4))
    for n, M in enumerate(R):
        tr = M[0, 0] + M[1, 1] + M[2, 2]
        if tr > 0:
            t = np.sqrt(tr + 1.0)
            w = 0.5 * t
            t = 0.5 / t
            v = [(M[2, 1] - M[1, 2]) * t, (M[0, 2] - M[2, 0]) * t, (M[1, 0] - M[0, 1]) * t, w]
        else:
            i = 0
            if M[1, 1] > M[0, 0]:
                i = 1
            if M[2, 2] > M[i, i]:
                i = 2
            j, k = (i + 1) % 3, (i + 2) % 3
            t = np.sqrt(M[i, i] - M[j, j] - M[k, k] + 1.0)
            v = [0.0] * 4
            v[i] = 0.5 * t
            t = 0.5 / t
            v[3] = (M[k, j] - M[j, k]) * t
            v[j] = (M[j, i] + M[i, j]) * t
            v[k] = (M[k, i] + M[i, k]) * t
        v = np.asarray(v) / np.linalg.norm(v)
        q[n] = -v if v[3] < 0 else v
    return q


def write_g2o_ba(path, prob):
    """openslam_g2o_amd.synthetic-style BA problem -> `.g2o` text (cameras get ids 0..P-1, points P..P+L-1)."""
    cams, pts = np.asarray(prob["cams"]), np.asarray(prob["pts"])
    P = len(cams)
    Rwc = cams[:, 0:9].reshape(-1, 3, 3).transpose(0, 2, 1)            # world -> camera, stored column-major
    twc = cams[:, 9:12]
    Rcw = Rwc.transpose(0, 2, 1)                                       # cam -> world
    tcw = -(Rcw @ twc[:, :, None])[:, :, 0]
    q = _R_to_quat(Rcw)
    with open(path, "w") as f:
        f.write("PARAMS_CAMERAPARAMETERS 0 %.17g %.17g %.17g 0\n" % (prob["f"], prob["cx"], prob["cy"]))
        for i in range(P):
            f.write("VERTEX_SE3:EXPMAP %d %s\n" % (i, " ".join("%.17g" % v for v in (*tcw[i], *q[i]))))
        for j in range(len(pts)):
            f.write("VERTEX_XYZ %d %s\n" % (P + j, " ".join("%.17g" % v for v in pts[j])))
        fixed = [i for i in range(P) if prob["cam_hidx"][i] < 0]
        if fixed:
            f.write("FIX %s\n" % " ".join(str(i) for i in fixed))
        info = prob.get("omega")
        for e in range(len(prob["meas"])):
            io = (1.0, 0.0, 1.0) if info is None else (info[e][0], info[e][2], info[e][3])   # column-major 2x2 -> upper
            f.write("EDGE_PROJECT_XYZ2UV:EXPMAP %d %d 0 %.17g %.17g %.17g %.17g %.17g\n" % (
                P + prob["pt_idx"][e], prob["cam_idx"][e], prob["meas"][e][0], prob["meas"][e][1], *io))


def read_g2o_ba(path):
    """`.g2o` BA file -> problem dict in the layout of openslam_g2o_amd.synthetic.make_ba_problem (index mapping:
    free poses by vertex id, then points by vertex id, sparse_optimizer.cpp:174-187; FIX or no FIX: gauge left to the caller)."""
    cam_id, cam_v, pt_id, pt_v, e_pt, e_cam, meas, info, fixed = [], [], [], [], [], [], [], [], []
    f_, cx, cy = None, None, None
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            tag = t[0]
            if tag == "PARAMS_CAMERAPARAMETERS":
                f_, cx, cy = float(t[2]), float(t[3]), float(t[4])
            elif tag == "VERTEX_SE3:EXPMAP":
                cam_id.append(int(t[1]))
                cam_v.append([float(x) for x in t[2:9]])
            elif tag in ("VERTEX_XYZ", "VERTEX_TRACKXYZ"):
                pt_id.append(int(t[1]))
                pt_v.append([float(x) for x in t[2:5]])
            elif tag == "EDGE_PROJECT_XYZ2UV:EXPMAP":
                e_pt.append(int(t[1]))
                e_cam.append(int(t[2]))
                meas.append([float(t[4]), float(t[5])])
                i00, i01, i11 = float(t[6]), float(t[7]), float(t[8])
                info.append([i00, i01, i01, i11])
            elif tag == "FIX":
                fixed.extend(int(x) for x in t[1:])
    if f_ is None:
        raise ValueError("no PARAMS_CAMERAPARAMETERS in %s" % path)
    cam_id, pt_id = np.asarray(cam_id, np.int64), np.asarray(pt_id, np.int64)
    oc, op = np.argsort(cam_id, kind="stable"), np.argsort(pt_id, kind="stable")
    cam_id, pt_id = cam_id[oc], pt_id[op]
    cv, pts = np.asarray(cam_v, np.float64)[oc], np.asarray(pt_v, np.float64)[op]
    Rcw = _quat_to_R(cv[:, 3:7])
    tcw = cv[:, 0:3]
    Rwc = Rcw.transpose(0, 2, 1)                                       # the vertex holds the inverse (world -> camera)
    twc = -(Rwc @ tcw[:, :, None])[:, :, 0]
    cams = np.empty((len(cv), 12))
    cams[:, 0:9] = Rwc.transpose(0, 2, 1).reshape(-1, 9)               # column-major
    cams[:, 9:12] = twc
    clut = {int(v): k for k, v in enumerate(cam_id)}
    plut = {int(v): k for k, v in enumerate(pt_id)}
    cam_idx = np.asarray([clut[a] for a in e_cam], np.int32)
    pt_idx = np.asarray([plut[a] for a in e_pt], np.int32)
    fx = {clut[v] for v in fixed if v in clut}
    cam_hidx = np.full(len(cams), -1, np.int32)
    k = 0
    for i in range(len(cams)):
        if i not in fx:
            cam_hidx[i] = k
            k += 1
    nP, L = k, len(pts)
    return dict(P=len(cams), L=L, E=len(meas), nP=nP, nL=L, f=f_, cx=cx, cy=cy, cams=cams, pts=pts,
                meas=np.asarray(meas, np.float64), omega=np.asarray(info, np.float64), cam_idx=cam_idx, pt_idx=pt_idx,
                cam_hidx=cam_hidx, v0=(nP + pt_idx).astype(np.int32), v1=cam_hidx[cam_idx].astype(np.int32))
