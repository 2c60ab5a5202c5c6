"""Synthetic Manhattan-world scans and voxel maps.

Deterministic numpy generators used by bench.py and the tests: there is no
network for datasets, and the reference publishes none.  The world is a ground
plane, a grid of box buildings, street poles and a perimeter wall; scans are
ray-cast with the reference's ring elevation tables
(/root/reference/L_SLAM/src/odometry/MultiScanRegistration.h:90-92 VLP-16,
:100-102 64-ring); the map is the set of voxel-spaced samples of the same
surfaces that a long accumulation of voxel-downsampled frames converges to
(corner leaf 0.2 m, surf leaf 0.4 m: ScanMatch.cpp:29-30).
"""
import numpy as np

SENSOR_HEIGHT = 1.8
MAX_RANGE = 140.0


def ring_elevations(rings):
    """MultiScanRegistration.h:90-92 (VLP-16: -15..+15 deg), :100-102 (64: -24.9..+2)."""
    if rings == 16:
        lo, hi = -15.0, 15.0
    elif rings == 64:
        lo, hi = -24.9, 2.0
    else:
        lo, hi = -15.0, 15.0
    return np.deg2rad(np.linspace(lo, hi, rings))


class World:
    """Axis-aligned boxes standing on z=0, inside a square perimeter wall."""

    def __init__(self, seed=20240601, half_extent=175.0, wall_half=90.0, wall_height=12.0,
                 pitch=40.0, pole_pitch=10.0):
        rng = np.random.default_rng(seed)
        self.half_extent = half_extent
        self.wall_half = wall_half
        self.wall_height = wall_height
        boxes = []
        poles = []
        n = int(np.floor(half_extent / pitch))
        for ix in range(-n, n + 1):
            for iy in range(-n, n + 1):
                cx, cy = ix * pitch + pitch / 2, iy * pitch + pitch / 2
                if abs(cx) > half_extent - 10 or abs(cy) > half_extent - 10:
                    continue
                sx, sy = rng.uniform(10, 22, 2)
                h = rng.uniform(6, 25)
                boxes.append((cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2, h))
        # street poles along the grid lines
        m = int(np.floor(half_extent / pole_pitch))
        for ix in range(-n, n + 1):
            for k in range(-m, m + 1):
                for (px, py) in ((ix * pitch + 2.0, k * pole_pitch + 1.0), (k * pole_pitch + 1.0, ix * pitch - 2.0)):
                    if abs(px) < half_extent - 5 and abs(py) < half_extent - 5:
                        poles.append((px - 0.12, px + 0.12, py - 0.12, py + 0.12, rng.uniform(4, 8)))
        self.boxes = np.array(boxes, np.float64)
        self.poles = np.array(poles, np.float64)
        w, t, hh = wall_half, 0.5, wall_height
        self.walls = np.array([(-w - t, -w, -w - t, w + t, hh), (w, w + t, -w - t, w + t, hh),
                               (-w - t, w + t, -w - t, -w, hh), (-w - t, w + t, w, w + t, hh)], np.float64)

    def solids(self):
        return np.concatenate([self.boxes, self.poles, self.walls], axis=0)


def _raycast(world, origin, dirs):
    """Nearest hit of each ray with ground / boxes.  Returns t, solid id (-1 ground), face axis."""
    o = origin[None, :]
    n = len(dirs)
    t_best = np.full(n, np.inf)
    sid = np.full(n, -2, np.int64)
    axis = np.zeros(n, np.int64)
    dz = dirs[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(dz < 0, -origin[2] / dz, np.inf)
    t_best = tg.copy()
    sid[np.isfinite(tg)] = -1
    axis[:] = 2
    solids = world.solids()
    inv = np.where(np.abs(dirs) > 1e-12, 1.0 / np.where(np.abs(dirs) > 1e-12, dirs, 1.0), np.inf)
    for k, (x0, x1, y0, y1, h) in enumerate(solids):
        lo = np.array([x0, y0, 0.0])
        hi = np.array([x1, y1, h])
        t1 = (lo[None, :] - o) * inv
        t2 = (hi[None, :] - o) * inv
        tmin = np.minimum(t1, t2)
        tmax = np.maximum(t1, t2)
        tn = tmin.max(axis=1)
        tf = tmax.min(axis=1)
        hit = (tn <= tf) & (tn > 1e-6) & (tn < t_best)
        if hit.any():
            t_best[hit] = tn[hit]
            sid[hit] = k
            axis[hit] = tmin[hit].argmax(axis=1)
    return t_best, sid, axis


def pose_to_Rt(pose):
    """R = Rz*Ry*Rx, p_map = R p + t (util/transform_utils.h:288-299), float64."""
    rx, ry, rz = pose[:3]
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx, np.asarray(pose[3:6], np.float64)


def make_scan(world, rings=64, azimuth_steps=1800, gt_pose=(0, 0, 0.3, 3.0, -2.0, SENSOR_HEIGHT),
              noise_sigma=0.02, seed=1234, corner_band=0.12, full=False):
    """Ray-cast one scan.  Returns (corner, surf) as (n,4) float32 {x,y,z,intensity} in the
    SENSOR frame (ring-major, azimuth-minor order) and the float64 ground-truth pose.
    intensity = ring id + relative time (util/pcl_util.h:30-37 semantics)."""
    rng = np.random.default_rng(seed)
    gt_pose = np.asarray(gt_pose, np.float64)
    R, t = pose_to_Rt(gt_pose)
    el = ring_elevations(rings)
    az = np.linspace(0, 2 * np.pi, azimuth_steps, endpoint=False)
    EL, AZ = np.meshgrid(el, az, indexing="ij")
    d_s = np.stack([np.cos(EL) * np.cos(AZ), np.cos(EL) * np.sin(AZ), np.sin(EL)], axis=-1).reshape(-1, 3)
    d_w = d_s @ R.T
    tt, sid, axis = _raycast(world, t, d_w)
    valid = np.isfinite(tt) & (tt < MAX_RANGE)
    rng_noise = rng.normal(0, noise_sigma, len(tt))
    r = np.where(valid, tt + rng_noise, 0.0)
    p_s = d_s * r[:, None]
    hit_w = t[None, :] + d_w * np.where(valid, tt, 0.0)[:, None]
    # corner label: hit on a pole, or on a box/wall face within corner_band of a vertical edge
    solids = world.solids()
    nb = len(world.boxes)
    npole = len(world.poles)
    is_corner = np.zeros(len(tt), bool)
    on_solid = sid >= 0
    s = solids[np.clip(sid, 0, len(solids) - 1)]
    is_pole = on_solid & (sid >= nb) & (sid < nb + npole)
    dx = np.minimum(np.abs(hit_w[:, 0] - s[:, 0]), np.abs(hit_w[:, 0] - s[:, 1]))
    dy = np.minimum(np.abs(hit_w[:, 1] - s[:, 2]), np.abs(hit_w[:, 1] - s[:, 3]))
    near_edge = on_solid & (axis != 2) & (np.where(axis == 0, dy, dx) < corner_band) & (sid < nb)
    is_corner = is_pole | near_edge
    ring = np.repeat(np.arange(rings), azimuth_steps)
    reltime = np.tile(np.arange(azimuth_steps) / azimuth_steps * 0.1, rings)
    pts = np.concatenate([p_s, (ring + reltime)[:, None]], axis=1).astype(np.float32)
    corner = pts[valid & is_corner]
    surf = pts[valid & ~is_corner]
    if full:
        # the ring-sorted full-resolution cloud MultiScanRegistration::process hands to
        # extractFeatures, with its per-ring [first, last] index ranges
        # (MultiScanRegistration.cpp:178-190); invalid returns are dropped like NaN points are
        cloud = pts[valid]
        cnt = np.bincount(ring[valid], minlength=rings)
        first = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        last = np.where(np.cumsum(cnt) > 0, np.cumsum(cnt) - 1, 0)
        ranges = np.stack([first, last], axis=1).astype(np.int32)
        return corner, surf, gt_pose, cloud, ranges
    return corner, surf, gt_pose


def _jittered_grid(rng, u0, u1, v0, v1, leaf):
    nu = max(1, int(np.floor((u1 - u0) / leaf)))
    nv = max(1, int(np.floor((v1 - v0) / leaf)))
    U, V = np.meshgrid(u0 + (np.arange(nu) + 0.5) * leaf, v0 + (np.arange(nv) + 0.5) * leaf, indexing="ij")
    U = U.ravel() + rng.uniform(-0.25, 0.25, U.size) * leaf
    V = V.ravel() + rng.uniform(-0.25, 0.25, V.size) * leaf
    return U, V


def make_map(world, leaf_corner=0.2, leaf_surf=0.4, radius=None, seed=77, sigma=0.01):
    """Voxel-spaced samples of every surface (surf map) and every vertical edge / pole
    (corner map) within `radius` (default: the whole world).  (n,4) float32, w = 0."""
    rng = np.random.default_rng(seed)
    H = world.half_extent if radius is None else radius
    surf = []
    corner = []
    # ground
    U, V = _jittered_grid(rng, -H, H, -H, H, leaf_surf)
    surf.append(np.stack([U, V, rng.normal(0, sigma, U.size)], 1))
    for group, is_pole in ((world.boxes, False), (world.walls, False), (world.poles, True)):
        for (x0, x1, y0, y1, h) in group:
            if min(abs(x0), abs(x1)) > H or min(abs(y0), abs(y1)) > H:
                continue
            if is_pole:
                z = np.arange(leaf_corner / 2, h, leaf_corner)
                cx, cy = (x0 + x1) / 2, (y0 + y1) / 2
                corner.append(np.stack([cx + rng.normal(0, sigma, z.size), cy + rng.normal(0, sigma, z.size),
                                        z + rng.uniform(-0.25, 0.25, z.size) * leaf_corner], 1))
                continue
            for (xa, xb, ya, yb) in ((x0, x0, y0, y1), (x1, x1, y0, y1), (x0, x1, y0, y0), (x0, x1, y1, y1)):
                if xa == xb:
                    U, V = _jittered_grid(rng, ya, yb, 0, h, leaf_surf)
                    surf.append(np.stack([xa + rng.normal(0, sigma, U.size), U, V], 1))
                else:
                    U, V = _jittered_grid(rng, xa, xb, 0, h, leaf_surf)
                    surf.append(np.stack([U, ya + rng.normal(0, sigma, U.size), V], 1))
            if group is world.boxes:
                z = np.arange(leaf_corner / 2, h, leaf_corner)
                for (ex, ey) in ((x0, y0), (x0, y1), (x1, y0), (x1, y1)):
                    corner.append(np.stack([ex + rng.normal(0, sigma, z.size), ey + rng.normal(0, sigma, z.size),
                                            z + rng.uniform(-0.25, 0.25, z.size) * leaf_corner], 1))
    surf = np.concatenate(surf, 0)
    corner = np.concatenate(corner, 0)
    keep_s = (np.abs(surf[:, 0]) <= H + 1) & (np.abs(surf[:, 1]) <= H + 1)
    keep_c = (np.abs(corner[:, 0]) <= H + 1) & (np.abs(corner[:, 1]) <= H + 1)
    surf, corner = surf[keep_s], corner[keep_c]
    # a long-running map has no spatial order: shuffle so kd-tree input order is generic
    surf = surf[rng.permutation(len(surf))]
    corner = corner[rng.permutation(len(corner))]
    pad = lambda a: np.concatenate([a, np.zeros((len(a), 1))], 1).astype(np.float32)
    return pad(corner), pad(surf)


def perturb_pose(gt_pose, seed=99, dt=0.3, dr_deg=2.0):
    """Initial guess: ground truth + uniform +-0.3 m / +-2 deg."""
    rng = np.random.default_rng(seed)
    p = np.asarray(gt_pose, np.float64).copy()
    p[:3] += np.deg2rad(rng.uniform(-dr_deg, dr_deg, 3))
    p[3:] += rng.uniform(-dt, dt, 3)
    return p.astype(np.float32)


def make_problem(rings=64, azimuth_steps=1800, map_radius=None, world_half=175.0, seed=0,
                 leaf_corner=0.2, leaf_surf=0.4):
    """World + map + one scan + perturbed initial pose."""
    world = World(half_extent=world_half, wall_half=min(90.0, world_half - 5.0))
    map_c, map_s = make_map(world, leaf_corner, leaf_surf, radius=map_radius, seed=77 + seed)
    gt = (0.01, -0.015, 0.3 + 0.1 * seed, 3.0 + seed, -2.0, SENSOR_HEIGHT)
    qc, qs, gt = make_scan(world, rings, azimuth_steps, gt_pose=gt, seed=1234 + seed)
    init = perturb_pose(gt, seed=99 + seed)
    return dict(world=world, map_corner=map_c, map_surf=map_s, corner=qc, surf=qs,
                gt_pose=gt.astype(np.float32), init_pose=init)


# ---------------------------------------------------------------------------
# synthetic stereo observations (BASELINE config 5: joint LiDAR + stereo system)
# ---------------------------------------------------------------------------
# camera looks along the LiDAR's +x: x_c = -y_l, y_c = -z_l, z_c = x_l, mounted 10 cm above / 5 cm ahead
T_CAM_LIDAR = np.array([[0.0, -1.0, 0.0, 0.0],
                        [0.0, 0.0, -1.0, 0.10],
                        [1.0, 0.0, 0.0, -0.05]])


def stereo_project(points_map, pose, fx=700.0, fy=700.0, cx=640.0, cy=360.0, bf=84.0, T_cl=T_CAM_LIDAR):
    """Map-frame points -> (uL, v, uR), camera depth; float64 (include/lslam_c.h conventions)."""
    R, t = pose_to_Rt(np.asarray(pose, np.float64))
    p = (np.asarray(points_map, np.float64)[:, :3] - t) @ R          # R^T (X - t)
    Xc = p @ T_cl[:, :3].T + T_cl[:, 3]
    z = Xc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        uL = fx * Xc[:, 0] / z + cx
        v = fy * Xc[:, 1] / z + cy
        uR = uL - bf / z
    return np.stack([uL, v, uR], 1), z


def make_stereo(points_map, gt_pose, n=2000, seed=5, sigma_px=0.5, mono_frac=0.1, outlier_frac=0.05,
                width=1280, height=720, max_depth=60.0, **cam):
    """Landmarks = map points in view of the camera at `gt_pose`; observations = their projections
    + Gaussian pixel noise scaled by an ORB pyramid level (1.2^level), some without a right match
    (uR = -1), some gross outliers.  -> landmarks (n,3) f32, obs (n,3) f32, inv_sigma2 (n,) f32."""
    rng = np.random.default_rng(seed)
    pts = np.asarray(points_map, np.float64)[:, :3]
    uvr, z = stereo_project(pts, gt_pose, **cam)
    ok = (z > 0.5) & (z < max_depth) & (uvr[:, 0] >= 0) & (uvr[:, 0] < width) & (uvr[:, 1] >= 0) & (uvr[:, 1] < height)
    cand = np.flatnonzero(ok)
    pick = cand[rng.permutation(len(cand))[:n]]
    lm, ob = pts[pick], uvr[pick]
    level = rng.integers(0, 8, len(pick))
    scale = 1.2 ** level
    ob = ob + rng.normal(0.0, sigma_px, ob.shape) * scale[:, None]
    out = rng.random(len(pick)) < outlier_frac
    ob[out, :2] += rng.uniform(-80.0, 80.0, (int(out.sum()), 2))
    mono = rng.random(len(pick)) < mono_frac
    ob[mono, 2] = -1.0
    return lm.astype(np.float32), ob.astype(np.float32), (1.0 / scale ** 2).astype(np.float32)


# ---------------------------------------------------------------------------
# synthetic pose graph (BASELINE config 4)
# ---------------------------------------------------------------------------
def _qmul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def _qrot(q, v):
    qv = np.concatenate([v, np.zeros(v.shape[:-1] + (1,))], axis=-1)
    qc = q * np.array([-1.0, -1.0, -1.0, 1.0])
    return _qmul(_qmul(q, qv), qc)[..., :3]


def _pmul(a, b):
    return np.concatenate([a[..., :3] + _qrot(a[..., 3:], b[..., :3]), _qmul(a[..., 3:], b[..., 3:])], axis=-1)


def _pinv(a):
    qi = a[..., 3:] * np.array([-1.0, -1.0, -1.0, 1.0])
    return np.concatenate([-_qrot(qi, a[..., :3]), qi], axis=-1)


def make_pose_graph(n_kf=5000, n_loop=20000, laps=8, radius=100.0, seed=7,
                    odo_sigma=(0.02, 0.002), loop_sigma=(0.01, 0.001)):
    """Keyframes {t, q_xyzw} on `laps` laps of a closed loop, n_kf-1 odometry edges with drift,
    n_loop loop-closure edges between keyframes < 5 m apart and > 30 m of path apart
    (pose_graph/loop_detector.hpp:57-60), information matrices as pose_graph/graph.cpp:279-288
    (odometry diag(0.8,0.4,0.8,1,2,1)) and :333-339 (loops 2*I).  Initial estimate: dead-reckoned."""
    rng = np.random.default_rng(seed)
    s = np.linspace(0, 2 * np.pi * laps, n_kf, endpoint=False)
    r = radius * (1 + 0.02 * np.sin(5 * s))
    pos = np.stack([r * np.cos(s), r * np.sin(s), 0.5 * np.sin(3 * s)], 1)
    yaw = s + np.pi / 2
    gt = np.concatenate([pos, np.stack([np.zeros(n_kf), np.zeros(n_kf), np.sin(yaw / 2), np.cos(yaw / 2)], 1)], 1)

    def noisy_rel(a, b, sig):
        rel = _pmul(_pinv(gt[a]), gt[b])
        v = rng.normal(0, sig[1], (len(a), 3))
        dq = np.concatenate([v, np.sqrt(1 - (v * v).sum(1, keepdims=True))], 1)
        d = np.concatenate([rng.normal(0, sig[0], (len(a), 3)), dq], 1)
        return _pmul(rel, d)

    a = np.arange(n_kf - 1)
    odo = noisy_rel(a, a + 1, odo_sigma)
    step = np.linalg.norm(pos[1] - pos[0])
    per_lap = n_kf // laps
    cand = np.zeros((0, 2), np.int64)
    while len(cand) < n_loop:  # same place, a whole number of laps later (+- a few keyframes)
        i = rng.integers(0, n_kf, 4 * n_loop)
        j = i + per_lap * rng.integers(1, laps, 4 * n_loop) + rng.integers(-3, 4, 4 * n_loop)
        ok = (j < n_kf) & (j > i)
        i, j = i[ok], j[ok]
        ok = ((j - i) * step > 30.0) & (np.linalg.norm(pos[i] - pos[j], axis=1) < 5.0)
        cand = np.concatenate([cand, np.stack([i[ok], j[ok]], 1)])
    cand = cand[:n_loop]
    loops = noisy_rel(cand[:, 0], cand[:, 1], loop_sigma)
    ij = np.concatenate([np.stack([a, a + 1], 1), cand]).astype(np.int32)
    meas = np.concatenate([odo, loops])
    info = np.zeros((len(ij), 6, 6))
    info[: n_kf - 1] = np.diag([0.8, 0.4, 0.8, 1.0, 2.0, 1.0])
    info[n_kf - 1:] = 2.0 * np.eye(6)
    init = np.zeros_like(gt)
    init[0] = gt[0]
    for k in range(n_kf - 1):
        init[k + 1] = _pmul(init[k][None], odo[k][None])[0]
    init[:, 3:] /= np.linalg.norm(init[:, 3:], axis=1, keepdims=True)
    return dict(gt=gt, init=init, ij=ij, meas=meas, info=info, n_odo=n_kf - 1)
