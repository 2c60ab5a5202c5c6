"""CPU ORACLE for the SE(3) pose-graph Levenberg-Marquardt (test infrastructure only).

PARITY UNPINNED.  The reference delegates this arithmetic to g2o
(/root/reference/L_SLAM/src/pose_graph/solver_g2o.cpp:5,16,51-95: VertexSE3, EdgeSE3,
algorithm "lm_var", csparse), which is not under /root/reference, is not pinned to a
version by the reference and is not installed here; the reference has no tests or
golden vectors for it.  This file restates g2o's published conventions in numpy fp64:

  * vertex estimate X_i in SE(3); update X <- X * fromVectorMQT(delta), delta =
    [dt(3), dq_xyz(3)], w = sqrt(1 - |dq|^2)           (g2o se3_ops / isometry3d_mappings)
  * edge error e = toVectorMQT(Z^-1 * X_i^-1 * X_j) = [t_e, q_e.xyz], q_e normalised with
    w >= 0                                              (g2o EdgeSE3::computeError)
  * chi2 = sum e^T Omega e; H = sum J^T Omega J; b = -sum J^T Omega e
  * Levenberg-Marquardt schedule of g2o's OptimizationAlgorithmLevenberg: lambda0 =
    1e-5 * max diag(H); rho = (chi2 - chi2_new) / (dx.(lambda dx + b) + 1e-3); accept:
    lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), ni = 2; reject: lambda *= ni, ni *= 2,
    at most 10 trials per iteration; first vertex fixed (solver_g2o.cpp:55-59)
  * information matrices as the reference builds them: odometry diag(0.8,0.4,0.8,1,2,1)
    (pose_graph/graph.cpp:279-288), loop closure 2*I (graph.cpp:333-339).

Jacobians here are taken by central differences of the error (independent of the analytic
ones in the HIP kernel); the linear system is solved with scipy's sparse LU.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def qmul(a, b):
    """Hamilton product, quaternions as (..., 4) arrays [x, y, z, w]."""
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def qconj(q):
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def qrot(q, v):
    """Rotate v by unit quaternion q."""
    qv = np.concatenate([v, np.zeros(v.shape[:-1] + (1,))], axis=-1)
    return qmul(qmul(q, qv), qconj(q))[..., :3]


def pose_mul(a, b):
    """SE(3) composition of poses [t(3), q(4)]."""
    return np.concatenate([a[..., :3] + qrot(a[..., 3:], b[..., :3]), qmul(a[..., 3:], b[..., 3:])], axis=-1)


def pose_inv(a):
    qi = qconj(a[..., 3:])
    return np.concatenate([-qrot(qi, a[..., :3]), qi], axis=-1)


def from_vector_mqt(d):
    """g2o fromVectorMQT: delta (..., 6) -> pose [t, q]."""
    v = d[..., 3:]
    w2 = 1.0 - (v * v).sum(-1, keepdims=True)
    w = np.sqrt(np.maximum(w2, 0.0))
    q = np.where(w2 < 0, np.array([0.0, 0.0, 0.0, 1.0]), np.concatenate([v, w], axis=-1))
    return np.concatenate([d[..., :3], q], axis=-1)


def edge_error(poses, ij, meas):
    """e = toVectorMQT(Z^-1 X_i^-1 X_j), shape (n_e, 6)."""
    xi, xj = poses[ij[:, 0]], poses[ij[:, 1]]
    E = pose_mul(pose_inv(meas), pose_mul(pose_inv(xi), xj))
    q = E[:, 3:]
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    q = np.where(q[:, 3:4] < 0, -q, q)
    return np.concatenate([E[:, :3], q[:, :3]], axis=1)


def chi2(poses, ij, meas, info):
    e = edge_error(poses, ij, meas)
    return float(np.einsum("ei,eij,ej->", e, info, e))


def oplus(poses, dx, fixed):
    """X_v <- X_v * fromVectorMQT(dx_v) for every free vertex."""
    d = dx.reshape(-1, 6).copy()
    d[fixed] = 0.0
    out = pose_mul(poses, from_vector_mqt(d))
    out[:, 3:] /= np.linalg.norm(out[:, 3:], axis=1, keepdims=True)
    return out


def numeric_jacobians(poses, ij, meas, h=1e-6):
    """Central differences of the edge error w.r.t. the local updates of vertex i and j."""
    ne = len(ij)
    Ji = np.zeros((ne, 6, 6))
    Jj = np.zeros((ne, 6, 6))
    for k in range(6):
        d = np.zeros((ne, 6))
        d[:, k] = h
        for sgn in (1.0, -1.0):
            pi = poses.copy()
            pi[ij[:, 0]] = pose_mul(poses[ij[:, 0]], from_vector_mqt(sgn * d))
            Ji[:, :, k] += sgn * _err_pairs(pi[ij[:, 0]], poses[ij[:, 1]], meas) / (2 * h)
            Jj[:, :, k] += sgn * _err_pairs(poses[ij[:, 0]], pose_mul(poses[ij[:, 1]], from_vector_mqt(sgn * d)), meas) / (2 * h)
    return Ji, Jj


def _err_pairs(xi, xj, meas):
    E = pose_mul(pose_inv(meas), pose_mul(pose_inv(xi), xj))
    q = E[:, 3:]
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    q = np.where(q[:, 3:4] < 0, -q, q)
    return np.concatenate([E[:, :3], q[:, :3]], axis=1)


def linearize(poses, ij, meas, info, e_begin=0, e_end=None):
    """Dense-block normal equations of the edges [e_begin, e_end):
    returns (H as scipy CSR 6n x 6n, b (6n), chi2)."""
    e_end = len(ij) if e_end is None else e_end
    sl = slice(e_begin, e_end)
    n = len(poses)
    ijs, ms, om = ij[sl], meas[sl], info[sl]
    e = edge_error(poses, ijs, ms)
    Ji, Jj = numeric_jacobians(poses, ijs, ms)
    Ai = np.einsum("eki,ekl->eil", Ji, om)  # Ji^T Omega
    Aj = np.einsum("eki,ekl->eil", Jj, om)
    Hii = np.einsum("eil,elj->eij", Ai, Ji)
    Hij = np.einsum("eil,elj->eij", Ai, Jj)
    Hjj = np.einsum("eil,elj->eij", Aj, Jj)
    bi = -np.einsum("eil,el->ei", Ai, e)
    bj = -np.einsum("eil,el->ei", Aj, e)
    rows, cols, vals = [], [], []
    r6 = np.arange(6)
    for (blk, a, c) in ((Hii, ijs[:, 0], ijs[:, 0]), (Hij, ijs[:, 0], ijs[:, 1]),
                        (np.transpose(Hij, (0, 2, 1)), ijs[:, 1], ijs[:, 0]), (Hjj, ijs[:, 1], ijs[:, 1])):
        rows.append((a[:, None, None] * 6 + r6[None, :, None] + 0 * r6[None, None, :]).ravel())
        cols.append((c[:, None, None] * 6 + 0 * r6[None, :, None] + r6[None, None, :]).ravel())
        vals.append(blk.ravel())
    H = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * n, 6 * n))
    b = np.zeros(6 * n)
    np.add.at(b, (ijs[:, 0, None] * 6 + r6[None, :]).ravel(), bi.ravel())
    np.add.at(b, (ijs[:, 1, None] * 6 + r6[None, :]).ravel(), bj.ravel())
    c2 = float(np.einsum("ei,eij,ej->", e, om, e))
    return H, b, c2


def solve_damped(H, b, lam, fixed):
    """(H + lam I) dx = b on the free vertices (fixed vertex removed, solver_g2o.cpp:55-59)."""
    n6 = H.shape[0]
    free = np.ones(n6, bool)
    free[fixed * 6: fixed * 6 + 6] = False
    idx = np.nonzero(free)[0]
    A = (H + lam * sp.identity(n6, format="csr"))[idx][:, idx].tocsc()
    dx = np.zeros(n6)
    dx[idx] = spla.spsolve(A, b[idx])
    return dx


def optimize(poses, ij, meas, info, fixed=0, max_iters=50, verbose=False):
    """g2o-style LM.  Returns (poses, history of dicts)."""
    poses = poses.copy()
    hist = []
    lam = None
    ni = 2.0
    for it in range(max_iters):
        H, b, cur = linearize(poses, ij, meas, info)
        if lam is None:
            d = H.diagonal().copy()
            d[fixed * 6: fixed * 6 + 6] = 0.0
            lam = 1e-5 * d.max()
        rho, qmax = 0.0, 0
        while True:
            dx = solve_damped(H, b, lam, fixed)
            trial = oplus(poses, dx, fixed)
            tmp = chi2(trial, ij, meas, info)
            scale = float(dx @ (lam * dx + b)) + 1e-3
            rho = (cur - tmp) / scale
            if rho > 0 and np.isfinite(tmp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                poses = trial
                cur_new = tmp
                accepted = True
            else:
                lam *= ni
                ni *= 2.0
                cur_new = cur
                accepted = False
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        hist.append(dict(iter=it, chi2=cur_new, lam=lam, trials=qmax, accepted=accepted))
        if verbose:
            print(hist[-1])
        if qmax == 10 or rho == 0:
            break
    return poses, hist


# ---------------------------------------------------------------------------
# synthetic pose graphs
# ---------------------------------------------------------------------------
def make_graph(n_kf=300, n_loop=1200, laps=3, radius=40.0, seed=7, odo_sigma=(0.02, 0.002), loop_sigma=(0.01, 0.001)):
    """Keyframes on `laps` laps of a closed loop (1 per ~step m), odometry edges with drift,
    loop edges between keyframes < 5 m apart and > 30 m of path apart (loop_detector.hpp:57-60),
    information matrices as graph.cpp:279-288,333-339.  Returns dict with ground truth."""
    rng = np.random.default_rng(seed)
    s = np.linspace(0, 2 * np.pi * laps, n_kf, endpoint=False)
    r = radius * (1 + 0.02 * np.sin(5 * s))
    pos = np.stack([r * np.cos(s), r * np.sin(s), 0.5 * np.sin(3 * s)], 1)
    yaw = s + np.pi / 2
    q = np.stack([np.zeros(n_kf), np.zeros(n_kf), np.sin(yaw / 2), np.cos(yaw / 2)], 1)
    gt = np.concatenate([pos, q], 1)

    def noisy_rel(a, b, sig):
        rel = pose_mul(pose_inv(gt[a]), gt[b])
        d = np.concatenate([rng.normal(0, sig[0], (len(a), 3)), rng.normal(0, sig[1], (len(a), 3))], 1)
        return pose_mul(rel, from_vector_mqt(d))

    a = np.arange(n_kf - 1)
    odo = noisy_rel(a, a + 1, odo_sigma)
    # loop candidates: close in space, far along the path
    step = np.linalg.norm(pos[1] - pos[0])
    cand = []
    tries = 0
    while len(cand) < n_loop and tries < 200 * n_loop:
        i = int(rng.integers(0, n_kf))
        j = int(rng.integers(0, n_kf))
        tries += 1
        if i < j and (j - i) * step > 30.0 and np.linalg.norm(pos[i] - pos[j]) < 5.0:
            cand.append((i, j))
    cand = np.array(cand, np.int64).reshape(-1, 2)
    loops = noisy_rel(cand[:, 0], cand[:, 1], loop_sigma)
    ij = np.concatenate([np.stack([a, a + 1], 1), cand]).astype(np.int32)
    meas = np.concatenate([odo, loops])
    info = np.zeros((len(ij), 6, 6))
    info[: n_kf - 1] = np.diag([0.8, 0.4, 0.8, 1.0, 2.0, 1.0])
    info[n_kf - 1:] = 2.0 * np.eye(6)
    # initial estimate: dead-reckoned odometry (drifts)
    init = np.zeros_like(gt)
    init[0] = gt[0]
    for k in range(n_kf - 1):
        init[k + 1] = pose_mul(init[k][None], odo[k][None])[0]
    init[:, 3:] /= np.linalg.norm(init[:, 3:], axis=1, keepdims=True)
    return dict(gt=gt, init=init, ij=ij, meas=meas, info=info, n_odo=n_kf - 1)
