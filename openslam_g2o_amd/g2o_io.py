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
