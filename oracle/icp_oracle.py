"""CPU ORACLE for the coarse loop-closure alignment (test infrastructure only).

PARITY UNPINNED.  The reference delegates ``LoopDetector::corseMatching``
(/root/reference/L_SLAM/src/pose_graph/loop_detector.hpp:61,232-255) to
``pcl::IterativeClosestPoint<PointXYZI, PointXYZI>`` with default settings; PCL is not under
/root/reference and not installed here, and the reference has no tests for it.  This file restates
PCL's published defaults independently of the device code -- scipy's cKDTree for the nearest neighbours,
numpy's SVD for the rigid fit:

  * correspondences: nearest target point of every transformed source point, all kept
    (max_correspondence_distance = sqrt(DBL_MAX));
  * TransformationEstimationSVD: R = V diag(1, 1, det(V U^T)) U^T from the SVD of the demeaned
    cross-covariance, t = c_t - R c_s; increments composed on the left;
  * DefaultConvergenceCriteria: 10 iterations at most (reaching them counts as converged), relative MSE
    change < 1e-5 or absolute < 1e-12; < 3 correspondences: not converged;
  * fitness = mean squared nearest-neighbour distance after the alignment.
"""
import numpy as np
from scipy.spatial import cKDTree


def icp_align(target, source, guess, max_iterations=10, transformation_epsilon=0.0, max_corr_dist=None):
    """-> (T 4x4 float64, converged, iterations, fitness)."""
    tgt = np.asarray(target, np.float64)[:, :3]
    src = np.asarray(source, np.float64)[:, :3]
    T = np.asarray(guess, np.float64).reshape(4, 4).copy()
    if len(tgt) == 0:
        return T, False, 0, 0.0
    tree = cKDTree(tgt)
    prev = np.finfo(np.float64).max
    converged, it = False, 0
    while True:
        cur = src @ T[:3, :3].T + T[:3, 3]
        d, idx = tree.query(cur)
        keep = np.ones(len(cur), bool) if max_corr_dist is None else d <= max_corr_dist
        if keep.sum() < 3:
            converged = False
            break
        s, t = cur[keep], tgt[idx[keep]]
        cs, ct = s.mean(0), t.mean(0)
        H = (s - cs).T @ (t - ct)
        U, _, Vt = np.linalg.svd(H)
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T)) or 1.0])
        R = Vt.T @ D @ U.T
        dt = ct - R @ cs
        inc = np.eye(4)
        inc[:3, :3], inc[:3, 3] = R, dt
        T = inc @ T
        it += 1
        mse = float((d[keep] ** 2).mean())
        if it >= max_iterations:
            converged = True
            break
        if 0.5 * (np.trace(R) - 1.0) >= 1.0 - transformation_epsilon and float(dt @ dt) <= transformation_epsilon:
            converged = True
            break
        if abs(mse - prev) < 1e-12 or abs(mse - prev) / prev < 1e-5:
            converged = True
            break
        prev = mse
    cur = src @ T[:3, :3].T + T[:3, 3]
    d, _ = tree.query(cur)
    return T, converged, it, float((d ** 2).mean()) if len(d) else 0.0
