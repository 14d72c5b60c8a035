"""Deterministic synthetic bundle-adjustment generator + host-side input producers.

The scene is the deterministic-window variant of g2o's `ba_demo`
(g2o/examples/ba/ba_demo.cpp:151-256) defined in SURVEY.md section 8d: focal length 1000,
principal point (320,240), information I_2; pose k sits at (k*s, 0, 0) with identity
rotation; landmark j is centred at pose c_j = floor(j*P/L), drawn uniformly in the box
x in t_c +- 1.5, y in +-0.5, z in [3,4] and observed by the 5 poses c_j-2..c_j+2 (window
shifted at the ends); pixel noise N(0,1); initial point perturbation N(0,0.05^2), pose
perturbation N(0,0.01^2) translation / N(0,0.005^2) rotation; poses 0 and 1 fixed
(ba_demo.cpp:182-184).  Randomness is a counter-based splitmix64 stream (seed 42) so the
same numbers come out of any vectorised or scalar implementation.

The error / Jacobian / oplus formulas are the host-side restatement of
EdgeProjectXYZ2UV (g2o/types/sba/types_six_dof_expmap.cpp:288-326, .h:139-147),
VertexSE3Expmap::oplusImpl (.h:101-104, SE3Quat::exp g2o/types/slam3d/se3quat.h:223-257)
and VertexSBAPointXYZ::oplusImpl (g2o/types/sba/types_sba.h:151-155): they are *input
producers* for the solver, not part of the accelerated path.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


class CounterRng:
    """u(stream, i): uniform in [0,1) from splitmix64(seed, stream, i)."""

    def __init__(self, seed=42):
        self.seed = np.uint64(seed)

    def uniform(self, stream, n):
        with np.errstate(over="ignore"):
            idx = np.arange(n, dtype=np.uint64)
            key = _splitmix64(self.seed * np.uint64(0x100000001B3) + np.uint64(stream))
            z = _splitmix64(key ^ (idx * np.uint64(0xD6E8FEB86659FD93) & _M64))
        return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, stream, n):
        u1 = self.uniform(2 * stream, n)
        u2 = self.uniform(2 * stream + 1, n)
        return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def _exp_so3(w):
    """Rodrigues as in SE3Quat::exp; w: [n,3] -> R [n,3,3], V [n,3,3]."""
    n = len(w)
    theta = np.linalg.norm(w, axis=1)
    Om = np.zeros((n, 3, 3))
    Om[:, 0, 1], Om[:, 0, 2] = -w[:, 2], w[:, 1]
    Om[:, 1, 0], Om[:, 1, 2] = w[:, 2], -w[:, 0]
    Om[:, 2, 0], Om[:, 2, 1] = -w[:, 1], w[:, 0]
    Om2 = Om @ Om
    I = np.eye(3)[None]
    small = theta < 0.00001
    th = np.where(small, 1.0, theta)
    a = (np.sin(th) / th)[:, None, None]
    b = ((1 - np.cos(th)) / (th * th))[:, None, None]
    c = ((th - np.sin(th)) / (th ** 3))[:, None, None]
    R = I + a * Om + b * Om2
    V = I + b * Om + c * Om2
    Rs = I + Om + Om2
    R = np.where(small[:, None, None], Rs, R)
    V = np.where(small[:, None, None], Rs, V)
    return R, V


def make_ba_problem(P, L, seed=42, spacing=0.5, obs_per_landmark=5, outlier_frac=0.0,
                    f=1000.0, cx=320.0, cy=240.0):
    rng = CounterRng(seed)
    K = obs_per_landmark
    assert P >= K
    j = np.arange(L, dtype=np.int64)
    c = (j * P) // L
    lo = np.clip(c - K // 2, 0, P - K)
    cam_idx = (lo[:, None] + np.arange(K)[None, :]).reshape(-1).astype(np.int32)       # landmark-major edges
    pt_idx = np.repeat(np.arange(L, dtype=np.int32), K)
    E = L * K
    true_pts = np.stack([c * spacing + (rng.uniform(1, L) * 3.0 - 1.5),
                         rng.uniform(2, L) - 0.5,
                         3.0 + rng.uniform(3, L)], axis=1)
    # true cameras: world->camera, R = I, t = -position
    cams_true = np.zeros((P, 12))
    cams_true[:, 0] = cams_true[:, 4] = cams_true[:, 8] = 1.0
    cams_true[:, 9] = -np.arange(P) * spacing
    # observations
    Xc = true_pts[pt_idx] + cams_true[cam_idx, 9:12]
    meas = np.stack([Xc[:, 0] / Xc[:, 2] * f + cx, Xc[:, 1] / Xc[:, 2] * f + cy], axis=1)
    meas[:, 0] += rng.normal(4, E)
    meas[:, 1] += rng.normal(5, E)
    if outlier_frac > 0:                      # ba_demo.cpp:222-227: uniform in the image
        is_out = rng.uniform(20, E) < outlier_frac
        meas[is_out, 0] = rng.uniform(21, E)[is_out] * 640.0
        meas[is_out, 1] = rng.uniform(22, E)[is_out] * 480.0
    # initial estimates
    pts = true_pts + 0.05 * np.stack([rng.normal(6, L), rng.normal(7, L), rng.normal(8, L)], axis=1)
    upd = np.zeros((P, 6))
    upd[:, 0:3] = 0.005 * np.stack([rng.normal(9, P), rng.normal(10, P), rng.normal(11, P)], axis=1)
    upd[:, 3:6] = 0.01 * np.stack([rng.normal(12, P), rng.normal(13, P), rng.normal(14, P)], axis=1)
    upd[:2] = 0.0                             # fixed poses keep the true value
    cams = _apply_cam_update(cams_true, upd)
    cam_hidx = np.arange(P, dtype=np.int32) - 2
    cam_hidx[:2] = -1
    nP = P - 2
    return dict(P=P, L=L, E=E, nP=nP, nL=L, f=f, cx=cx, cy=cy, cams=cams, pts=pts, meas=meas,
                cam_idx=cam_idx, pt_idx=pt_idx, cam_hidx=cam_hidx,
                v0=(nP + pt_idx).astype(np.int32),        # EdgeProjectXYZ2UV: vertex 0 = point
                v1=cam_hidx[cam_idx].astype(np.int32))    # vertex 1 = pose (types_six_dof_expmap.h:133)


def make_ba_loops(P, L, laps=4, hubs=3, drop=0.2, seed=7, spacing=0.5, f=1000.0, cx=320.0, cy=240.0, hub_stride=3):
    """A bundle-adjustment graph that is NOT a band: the camera runs `laps` times back and forth along the same line
    (poses of different laps stand at the same places and see the same points -- loop closures everywhere, the reduced
    pose system couples every lap with every other: separators and frontal matrices several times the band case), a
    fraction `drop` of the observations is missing (ragged observation lists, 2 .. 5 laps long) and `hubs` distant points
    are seen by every third pose (`hub_stride`; lists far longer than a wavefront).  Same dictionary as make_ba_problem."""
    rng = CounterRng(seed)
    M = max(P // laps, 6)                                   # positions along the line
    c = np.arange(P, dtype=np.int64)
    lap = c // M
    s_c = np.where(lap % 2 == 0, c % M, M - 1 - (c % M))  # position index of pose c
    pos_of = [np.flatnonzero(s_c == s) for s in range(M)]  # poses standing at position s
    Lr = L - hubs
    j = np.arange(Lr, dtype=np.int64)
    p_j = (j * M) // max(Lr, 1)
    cam_l, pt_l = [], []
    keep_u = rng.uniform(30, Lr * 5 * (laps + 1)).reshape(Lr, -1)
    for jj in range(Lr):
        obs = np.concatenate([pos_of[s] for s in range(max(p_j[jj] - 2, 0), min(p_j[jj] + 3, M))])
        obs.sort()
        k = keep_u[jj, :len(obs)] >= drop
        if k.sum() < 2:
            k[:2] = True
        obs = obs[k]
        cam_l.append(obs)
        pt_l.append(np.full(len(obs), jj))
    for h in range(hubs):
        obs = np.arange(h, P, hub_stride)
        cam_l.append(obs)
        pt_l.append(np.full(len(obs), Lr + h))
    cam_idx = np.concatenate(cam_l).astype(np.int32)
    pt_idx = np.concatenate(pt_l).astype(np.int32)
    E = len(cam_idx)
    true_pts = np.zeros((L, 3))
    true_pts[:Lr] = np.stack([p_j * spacing + (rng.uniform(1, Lr) * 2.0 - 1.0), rng.uniform(2, Lr) - 0.5, 3.0 + rng.uniform(3, Lr)], axis=1)
    for h in range(hubs):
        true_pts[Lr + h] = [0.5 * M * spacing + 3.0 * (h - hubs / 2.0), 1.0 * h, 60.0 + 5.0 * h]
    cams_true = np.zeros((P, 12))
    cams_true[:, 0] = cams_true[:, 4] = cams_true[:, 8] = 1.0
    cams_true[:, 9] = -s_c * spacing
    cams_true[:, 10] = -0.03 * lap
    Xc = true_pts[pt_idx] + cams_true[cam_idx, 9:12]
    meas = np.stack([Xc[:, 0] / Xc[:, 2] * f + cx, Xc[:, 1] / Xc[:, 2] * f + cy], axis=1)
    meas[:, 0] += rng.normal(4, E)
    meas[:, 1] += rng.normal(5, E)
    pts = true_pts + 0.05 * np.stack([rng.normal(6, L), rng.normal(7, L), rng.normal(8, L)], axis=1)
    upd = np.zeros((P, 6))
    upd[:, 0:3] = 0.005 * np.stack([rng.normal(9, P), rng.normal(10, P), rng.normal(11, P)], axis=1)
    upd[:, 3:6] = 0.01 * np.stack([rng.normal(12, P), rng.normal(13, P), rng.normal(14, P)], axis=1)
    upd[:2] = 0.0
    cams = _apply_cam_update(cams_true, upd)
    cam_hidx = np.arange(P, dtype=np.int32) - 2
    cam_hidx[:2] = -1
    nP = P - 2
    return dict(P=P, L=L, E=E, nP=nP, nL=L, f=f, cx=cx, cy=cy, cams=cams, pts=pts, meas=meas,
                cam_idx=cam_idx, pt_idx=pt_idx, cam_hidx=cam_hidx,
                v0=(nP + pt_idx).astype(np.int32), v1=cam_hidx[cam_idx].astype(np.int32))


def make_ba_grid(P, pts_per_cam=10, radius=0.9, seed=11, spacing=0.5, f=1000.0, cx=320.0, cy=240.0):
    """A bundle-adjustment graph with VISIBILITY BY DISTANCE (round 6; the shape of an aerial survey / a street grid seen from above,
    after g2o's ba_demo geometry, g2o/examples/ba/ba_demo.cpp:151-256, and the visibility lists of a BAL file,
    g2o/examples/bal/bal_example.cpp:294-406): G x G cameras stand on a square lattice (pitch `spacing`) in the plane z = 0 and
    look along +z, `pts_per_cam` points per camera are spread uniformly over the area at depth 3 .. 4, and a point is observed by
    EVERY camera within the horizontal distance `radius` of it (6 .. 13 observations per point at the defaults: ragged lists).  Two
    cameras share points when they are closer than 2 x radius: every camera is coupled to ~40 others, the reduced pose system is
    a two-dimensional mesh with a wide stencil -- real fill, separators of hundreds of columns, frontal matrices far beyond the
    24-column register fronts of the camera-trajectory graphs.  P is rounded down to a square.  Same dictionary as
    make_ba_problem (edges point-major, cameras row-major on the lattice, the first two cameras fixed)."""
    rng = CounterRng(seed)
    G = int(np.floor(np.sqrt(P)))
    P = G * G
    L = P * pts_per_cam
    side = (G - 1) * spacing
    px = rng.uniform(1, L) * side
    py = rng.uniform(2, L) * side
    true_pts = np.stack([px, py, 3.0 + rng.uniform(3, L)], axis=1)
    reach = int(np.ceil(radius / spacing))
    off = np.arange(-reach, reach + 1)
    oi, oj = np.meshgrid(off, off, indexing="ij")
    oi, oj = oi.reshape(-1), oj.reshape(-1)
    ci = np.rint(px / spacing).astype(np.int64)[:, None] + oi[None, :]
    cj = np.rint(py / spacing).astype(np.int64)[:, None] + oj[None, :]
    d2 = (ci * spacing - px[:, None]) ** 2 + (cj * spacing - py[:, None]) ** 2
    vis = (ci >= 0) & (ci < G) & (cj >= 0) & (cj < G) & (d2 <= radius * radius)
    cam_all = (ci * G + cj)
    # point-major edge list, cameras ascending inside a list
    order = np.argsort(np.where(vis, cam_all, np.iinfo(np.int64).max), axis=1, kind="stable")
    cam_sorted = np.take_along_axis(cam_all, order, axis=1)
    vis_sorted = np.take_along_axis(vis, order, axis=1)
    cam_idx = cam_sorted[vis_sorted].astype(np.int32)
    pt_idx = np.repeat(np.arange(L, dtype=np.int32), vis_sorted.sum(axis=1))
    assert vis_sorted.sum(axis=1).min() >= 2, "a point with fewer than two observations"
    E = len(cam_idx)
    cams_true = np.zeros((P, 12))
    cams_true[:, 0] = cams_true[:, 4] = cams_true[:, 8] = 1.0
    cams_true[:, 9] = -(np.arange(P) // G) * spacing
    cams_true[:, 10] = -(np.arange(P) % G) * spacing
    Xc = true_pts[pt_idx] + cams_true[cam_idx, 9:12]
    meas = np.stack([Xc[:, 0] / Xc[:, 2] * f + cx, Xc[:, 1] / Xc[:, 2] * f + cy], axis=1)
    meas[:, 0] += rng.normal(4, E)
    meas[:, 1] += rng.normal(5, E)
    pts = true_pts + 0.05 * np.stack([rng.normal(6, L), rng.normal(7, L), rng.normal(8, L)], axis=1)
    upd = np.zeros((P, 6))
    upd[:, 0:3] = 0.005 * np.stack([rng.normal(9, P), rng.normal(10, P), rng.normal(11, P)], axis=1)
    upd[:, 3:6] = 0.01 * np.stack([rng.normal(12, P), rng.normal(13, P), rng.normal(14, P)], axis=1)
    upd[:2] = 0.0
    cams = _apply_cam_update(cams_true, upd)
    cam_hidx = np.arange(P, dtype=np.int32) - 2
    cam_hidx[:2] = -1
    nP = P - 2
    return dict(P=P, L=L, E=E, nP=nP, nL=L, f=f, cx=cx, cy=cy, cams=cams, pts=pts, meas=meas,
                cam_idx=cam_idx, pt_idx=pt_idx, cam_hidx=cam_hidx,
                v0=(nP + pt_idx).astype(np.int32), v1=cam_hidx[cam_idx].astype(np.int32))


def _apply_cam_update(cams, upd):
    """estimate <- exp(update) * estimate, update = (omega, upsilon)."""
    R, V = _exp_so3(upd[:, 0:3])
    Rold = cams[:, 0:9].reshape(-1, 3, 3).transpose(0, 2, 1)       # stored column-major
    told = cams[:, 9:12]
    Rn = R @ Rold
    tn = (R @ told[:, :, None])[:, :, 0] + (V @ upd[:, 3:6, None])[:, :, 0]
    out = np.empty_like(cams)
    out[:, 0:9] = Rn.transpose(0, 2, 1).reshape(-1, 9)
    out[:, 9:12] = tn
    return out


def ba_linearize(prob, jac=True):
    """Returns (Jpt [E,6] 2x3 col-major, Jcam [E,12] 2x6 col-major, err [E,2]) or err only."""
    cams, pts = prob["cams"], prob["pts"]
    f, cx, cy = prob["f"], prob["cx"], prob["cy"]
    T = cams[prob["cam_idx"]]
    X = pts[prob["pt_idx"]]
    R = T[:, 0:9].reshape(-1, 3, 3).transpose(0, 2, 1)
    Xc = (R @ X[:, :, None])[:, :, 0] + T[:, 9:12]
    x, y, z = Xc[:, 0], Xc[:, 1], Xc[:, 2]
    err = np.stack([prob["meas"][:, 0] - (x / z * f + cx), prob["meas"][:, 1] - (y / z * f + cy)], axis=1)
    if not jac:
        return err
    E = len(x)
    z2 = z * z
    tmp = np.zeros((E, 2, 3))
    tmp[:, 0, 0] = f
    tmp[:, 0, 2] = -x / z * f
    tmp[:, 1, 1] = f
    tmp[:, 1, 2] = -y / z * f
    A = (-1.0 / z)[:, None, None] * (tmp @ R)
    Jpt = np.ascontiguousarray(A.transpose(0, 2, 1).reshape(E, 6))          # column-major 2x3
    B = np.zeros((E, 2, 6))
    B[:, 0, 0] = x * y / z2 * f
    B[:, 0, 1] = -(1 + (x * x / z2)) * f
    B[:, 0, 2] = y / z * f
    B[:, 0, 3] = -1.0 / z * f
    B[:, 0, 5] = x / z2 * f
    B[:, 1, 0] = (1 + y * y / z2) * f
    B[:, 1, 1] = -x * y / z2 * f
    B[:, 1, 2] = -x / z * f
    B[:, 1, 4] = -1.0 / z * f
    B[:, 1, 5] = y / z2 * f
    Jcam = np.ascontiguousarray(B.transpose(0, 2, 1).reshape(E, 12))
    return Jpt, Jcam, err


def ba_omega(prob):
    om = np.zeros((prob["E"], 4))
    om[:, 0] = om[:, 3] = 1.0
    return om


def ba_oplus(prob, x):
    """Apply a solver update (poses then landmarks) and return a new problem dict."""
    nP = prob["nP"]
    xp = x[:6 * nP].reshape(nP, 6)
    xl = x[6 * nP:].reshape(prob["nL"], 3)
    upd = np.zeros((prob["P"], 6))
    free = prob["cam_hidx"] >= 0
    upd[free] = xp[prob["cam_hidx"][free]]
    new = dict(prob)
    new["cams"] = _apply_cam_update(prob["cams"], upd)
    new["pts"] = prob["pts"] + xl
    return new


def _quat_to_rot(q):
    """Unit quaternions [n,4] (x, y, z, w) -> rotation matrices [n,3,3]."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _iso_pack(R, t):
    """[n,3,3], [n,3] -> [n,12] (R column-major | t): the isometry layout of the device front ends and the oracle."""
    return np.concatenate([R.transpose(0, 2, 1).reshape(-1, 9), t], axis=1)


def make_sphere(nodes_per_level=50, laps=50, radius=100.0, noise_translation=0.01, noise_rotation=0.005, seed=42):
    """The generator of BASELINE.json's config 2 in its stated size: g2o's `create_sphere` with its defaults
    (g2o/examples/sphere/create_sphere.cpp:55-57: 50 nodes per lap x 50 laps = 2 500 VertexSE3 on a sphere of radius 100;
    :99-147 odometry edges between successive vertices and three loop closures per vertex to the lap before; :149-176
    measurement noise sigma_t = 0.01, sigma_r = 0.005 on the quaternion's vector part with w = 1 - |v| before the
    normalisation; :178-185 initial estimates by chaining the NOISY odometry), with this module's counter-based RNG instead
    of the reference's unseeded sampler.  Vertex 0 is fixed.  Returns isometries as [n,12] (R column-major | t)."""
    n = nodes_per_level * laps
    idx = np.arange(n)
    nn, f = idx % nodes_per_level, idx // nodes_per_level
    az = -np.pi + 2.0 * nn * np.pi / nodes_per_level
    ay = -0.5 * np.pi + (idx + 1) * np.pi / n                 # (the reference uses the post-incremented id here)
    cz, sz, cy, sy = np.cos(az), np.sin(az), np.cos(ay), np.sin(ay)
    Rz = np.zeros((n, 3, 3)); Ry = np.zeros((n, 3, 3))
    Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = cz, -sz, sz, cz, 1.0
    Ry[:, 0, 0], Ry[:, 0, 2], Ry[:, 2, 0], Ry[:, 2, 2], Ry[:, 1, 1] = cy, sy, -sy, cy, 1.0
    R = Rz @ Ry
    t = R[:, :, 0] * radius
    vi = [np.arange(n - 1)]
    vj = [np.arange(1, n)]
    for lap in range(1, laps):
        for d in (-1, 0, 1):
            a = (lap - 1) * nodes_per_level + np.arange(nodes_per_level)
            b = lap * nodes_per_level + np.arange(nodes_per_level) + d
            if lap == laps - 1 and d == 1:
                continue
            vi.append(a)
            vj.append(b)
    # (the reference emits the three closures of a vertex together; the order of the edges only matters for the noise stream)
    vi, vj = np.concatenate(vi).astype(np.int32), np.concatenate(vj).astype(np.int32)
    E = len(vi)
    Rt = R.transpose(0, 2, 1)
    Rm = Rt[vi] @ R[vj]                                          # from^-1 * to
    tm = np.einsum("nij,nj->ni", Rt[vi], t[vj] - t[vi])
    rng = CounterRng(seed)
    qv = np.stack([rng.normal(100 + c, E) for c in range(3)], axis=1) * noise_rotation
    qw = np.maximum(1.0 - np.linalg.norm(qv, axis=1), 0.0)
    q = np.concatenate([qv, qw[:, None]], axis=1)
    q /= np.linalg.norm(q, axis=1)[:, None]
    Rm = Rm @ _quat_to_rot(q)
    tm = tm + np.stack([rng.normal(110 + c, E) for c in range(3)], axis=1) * noise_translation
    # initial estimates: vertex 0 at its true pose, the others by chaining the noisy odometry (EdgeSE3::initialEstimate)
    Re, te = np.empty_like(R), np.empty_like(t)
    Re[0], te[0] = R[0], t[0]
    for i in range(1, n):
        Re[i] = Re[i - 1] @ Rm[i - 1]
        te[i] = Re[i - 1] @ tm[i - 1] + te[i - 1]
    info = np.zeros((6, 6))
    info[:3, :3] = np.eye(3) / noise_translation ** 2
    info[3:, 3:] = np.eye(3) / noise_rotation ** 2
    hidx = np.arange(n, dtype=np.int32) - 1                      # vertex 0 fixed
    return dict(n=n, nP=n - 1, E=E, vi=vi, vj=vj, poses=_iso_pack(Re, te), poses_true=_iso_pack(R, t), Z=_iso_pack(Rm, tm),
                omega=np.tile(info.T.reshape(1, 36), (E, 1)), hidx=hidx)
