"""Host-side mirror of the loop-closure front end, ``pose_graph::LoopDetector``
(/root/reference/L_SLAM/src/pose_graph/loop_detector.hpp:50-280): trajectory radius search,
candidate gating, coarse alignment hook, ``ScanMatch::scanMatchLocal`` fine alignment (SURVEY 8f row
n3).  The arithmetic that matters -- the fine alignment -- runs on the device through
:class:`ScanMatch`; the gating is a few comparisons per keyframe and stays on the host, as in the
reference.

Three things of the reference are restated as they are, not as they were probably meant:
  * ``radiusSearch(pos, 5.0, ...)`` (loop_detector.hpp:124-126) hands 5.0 to nanoflann's
    RadiusResultSet, which compares SQUARED distances with it (util/nanoflann_pcl.h:166-186), so
    the search radius is sqrt(5) m and the ``>= estimated_distance_thresh`` break (:133) never fires;
  * the same ``KdTreeFLANN::radiusSearch`` takes its result count from
    ``_kdtree.findNeighbors(...)`` (nanoflann_pcl.h:173), which in the vendored nanoflann returns
    ``result.full()`` -- a bool, always true for a RadiusResultSet (nanoflann.hpp:171,1304-1322) --
    so it hands back exactly ONE result, the nearest trajectory point (after the sort, :175-176).
    Checked against the reference's own nanoflann (tests/test_loop_closure.py).  When nothing lies
    within the radius the reference still reports one result and reads element 0 of an empty
    vector (undefined behaviour: in practice a stale entry of an earlier search); that is not
    reproduced -- ``radius_search`` returns nothing then.  ``single_result_quirk = False`` gives
    the search its documented meaning (all points within the radius, ascending);
  * the trajectory is flattened with ``pos.y = 0`` (:98,120).
The coarse alignment (``corseMatching``, :232-255) is PCL's ``IterativeClosestPoint`` with default
settings -- external, parity unpinned: the default ``coarse_matcher`` is the device ICP
(``lslam_icp_align``, csrc/lslam_icp.hip, PCL's defaults restated), including the reference's two
guards -- an empty reference cloud and a non-converged ICP both reject the candidate.  Any
``(reference_surf, surf, guess4x4) -> (converged, guess4x4)`` callable can be passed instead.
"""
import numpy as np

from .scan_match import ScanMatch


class KeyFrame:
    """pose_graph/keyframe.h: estimate (4x4, float64), accumulated travel distance, feature clouds
    ((n, 4) {x,y,z,intensity} in the keyframe's own frame)."""

    def __init__(self, estimate, accum_distance, corner_cloud, surf_cloud, frame_id=0):
        self.estimate = np.asarray(estimate, np.float64).reshape(4, 4)
        self.accum_distance = float(accum_distance)
        self.corner_cloud = np.ascontiguousarray(corner_cloud, np.float32)
        self.surf_cloud = np.ascontiguousarray(surf_cloud, np.float32)
        self.frame_id = frame_id


class Loop:
    """loop_detector.hpp:18-48."""

    def __init__(self, key1, key2, relative_pose):
        self.key1, self.key2 = key1, key2
        self.relative_pose = np.asarray(relative_pose, np.float32).reshape(4, 4)


def transform_cloud(cloud, tf):
    """pcl::transformPointCloud with an Isometry3f: p' = R p + t, fp32, intensity kept."""
    c = np.ascontiguousarray(cloud, np.float32)
    T = np.asarray(tf, np.float32)
    out = c.copy()
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    for r in range(3):
        out[:, r] = ((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3]
    return out


class LoopDetector:
    def __init__(self, scan_match=None, coarse_matcher=None, device=0, ctx=None):
        # loop_detector.hpp:55-63
        self.single_result_quirk = True  # nanoflann_pcl.h:173 (see the module docstring)
        self.estimated_distance_thresh = 25.0
        self.accum_distance_thresh = 30.0
        self.last_loop_interval_thresh = 3.0
        self.fitness_score_thresh = 0.5
        self.loop_count = 0
        self.last_loop_accum_distance = 0.0
        self._trajectory = np.zeros((0, 4), np.float32)
        self._scan_match = scan_match
        self._device, self._ctx = device, ctx
        self.coarse_matcher = coarse_matcher or self._icp_coarse_matcher

    @property
    def scan_match(self):
        if self._scan_match is None:  # created on first use: the gating needs no device
            self._scan_match = ScanMatch(10, device=self._device, ctx=self._ctx)
        return self._scan_match

    def _icp_coarse_matcher(self, refer_surf, surf, guess):
        """corseMatching, loop_detector.hpp:232-255: registration->align(aligned, guess) with PCL's defaults."""
        if len(refer_surf) == 0:  # :233-235
            return False, guess
        T, converged, _its, _fit = self.scan_match.ctx.icp_align(refer_surf, surf, guess)
        return converged, T

    def get_distance_thresh(self):
        return self.estimated_distance_thresh

    def get_loop_count(self):
        return self.loop_count

    # ---- loop_detector.hpp:66-87 --------------------------------------------------------------
    def detect_nearest(self, keyframes, new_keyframes):
        """-> list of Loop (the reference's bool is ``len(result) > 0``)."""
        self.update_trajectory(keyframes)
        loops = []
        for nk in new_keyframes:
            cand = self.find_nearest_candidates(keyframes, nk)
            if cand:
                loop = self.matching_nearest(cand, nk)
                if loop is not None:
                    loops.append(loop)
                    self.loop_count += 1
        return loops

    # ---- :93-106 ----------------------------------------------------------------------------------
    def update_trajectory(self, keyframes):
        t = np.zeros((len(keyframes), 4), np.float32)
        for i, k in enumerate(keyframes):
            t[i, :3] = k.estimate[:3, 3].astype(np.float32)
            t[i, 1] = 0.0
            t[i, 3] = i
        self._trajectory = t

    def radius_search(self, pos, radius):
        """KdTreeFLANN::radiusSearch (util/nanoflann_pcl.h:166-186): `radius` is compared with
        SQUARED distances; results ascending by distance."""
        t = self._trajectory
        d = t[:, :3] - np.asarray(pos, np.float32)[None, :3]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        idx = np.nonzero(d2 < np.float32(radius))[0]
        order = np.argsort(d2[idx], kind="stable")
        if self.single_result_quirk:
            order = order[:1]
        return idx[order], d2[idx][order]

    # ---- :108-164 ---------------------------------------------------------------------------------
    def find_nearest_candidates(self, keyframes, new_keyframe):
        if new_keyframe.accum_distance - self.last_loop_accum_distance < self.last_loop_interval_thresh:
            return []
        pos = new_keyframe.estimate[:3, 3].astype(np.float32).copy()
        pos[1] = 0.0
        r_indices, r_sqr = self.radius_search(pos, 5.0)
        if len(r_indices) <= 0:
            return []
        candidates = []
        candidate_dist = 0.0
        for i in range(len(r_sqr)):
            if len(candidates) >= 6:
                break
            if r_sqr[i] >= self.estimated_distance_thresh:
                break
            keyframe_index = int(round(float(self._trajectory[r_indices[i], 3])))
            kf = keyframes[keyframe_index]
            if new_keyframe.accum_distance - kf.accum_distance < self.accum_distance_thresh:
                continue
            if not candidates:
                candidate_dist = kf.accum_distance
            elif abs(candidate_dist - kf.accum_distance) > 5.0:
                continue
            candidates.append(kf)
        return candidates

    # ---- :166-230 ---------------------------------------------------------------------------------
    def matching_nearest(self, candidate_keyframes, new_keyframe):
        if not candidate_keyframes:
            return None
        candidate_tf = candidate_keyframes[0].estimate
        inv = np.linalg.inv(candidate_tf)
        corner_local = [candidate_keyframes[0].corner_cloud]
        surf_local = [candidate_keyframes[0].surf_cloud]
        for k in candidate_keyframes[1:]:
            rel = (inv @ k.estimate).astype(np.float32)
            corner_local.append(transform_cloud(k.corner_cloud, rel))
            surf_local.append(transform_cloud(k.surf_cloud, rel))
        corner_local = np.concatenate(corner_local)
        surf_local = np.concatenate(surf_local)
        new_relative = (inv @ new_keyframe.estimate).astype(np.float32)
        ok, coarse = self.coarse_matcher(surf_local, new_keyframe.surf_cloud, new_relative.copy())
        if not ok:
            return None
        converged, guess2 = self.scan_match.scanMatchLocal(corner_local, surf_local, new_keyframe.corner_cloud,
                                                           new_keyframe.surf_cloud,
                                                           np.asarray(coarse, np.float32).reshape(4, 4))
        if not converged:
            return None
        self.last_loop_accum_distance = new_keyframe.accum_distance
        return Loop(candidate_keyframes[0], new_keyframe, guess2)
