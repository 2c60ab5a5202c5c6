"""Host-side mirror of the pose-graph node's bookkeeping, ``pose_graph::Graph`` +
``KeyframeUpdater`` (/root/reference/L_SLAM/src/pose_graph/graph.cpp:230-385,
keyframe_updater.hpp:10-60): keyframe selection, odometry edges with the reference's information
matrices, loop detection, optimisation and the odom -> graph correction.  The
ROS plumbing (topics, threads, tf) is not mirrored; the arithmetic runs on the device through
:class:`PoseGraph` (LM) and :class:`LoopDetector` (scanMatchLocal).
"""
import numpy as np

from .loop_closure import KeyFrame, LoopDetector
from .pose_graph import PoseGraph


def _angle_of(R):
    """Eigen::AngleAxisd(R).angle() in [0, pi]."""
    c = (np.trace(R) - 1.0) / 2.0
    return float(np.arccos(min(1.0, max(-1.0, c))))


class KeyframeUpdater:
    """keyframe_updater.hpp:10-60."""

    def __init__(self):
        self.is_first = True
        self.prev_keypose = np.eye(4)
        self.keyframe_delta_trans = 0.25
        self.keyframe_delta_angle = 0.05
        self.accum_distance = 0.0
        self.frame_count = 0
        self.frame_id = 0

    def update(self, pose):
        pose = np.asarray(pose, np.float64).reshape(4, 4)
        if self.is_first:
            self.is_first = False
            self.prev_keypose = pose.copy()
            return True
        delta = np.linalg.inv(self.prev_keypose) @ pose
        dx = float(np.linalg.norm(delta[:3, 3]))
        da = _angle_of(delta[:3, :3])
        if dx < self.keyframe_delta_trans and da < self.keyframe_delta_angle:
            return False
        self.accum_distance += dx
        self.prev_keypose = pose.copy()
        self.frame_count += 1
        return True

    def get_accum_distance(self):
        return self.accum_distance

    def get_unique_id(self):
        self.frame_id += 1
        return self.frame_id


# graph.cpp:279-288 and :333-339
ODOMETRY_INFORMATION = np.diag([0.8, 0.4, 0.8, 1.0, 2.0, 1.0])
LOOP_INFORMATION = 2.0 * np.eye(6)


class Graph:
    def __init__(self, device=0, ctx=None, loop_detector=None, max_keyframes_per_update=10):
        self.solver = PoseGraph(device)
        self.loop_detector = loop_detector or LoopDetector(device=device, ctx=ctx)
        self.keyframe_updater = KeyframeUpdater()
        self.keyframes = []
        self.new_keyframes = []
        self.keyframe_queue = []
        self.max_keyframes_per_update = max_keyframes_per_update
        self.tf_odom2graph = np.eye(4)
        self.loops = []

    # graph.cpp:230-246
    def add_frame(self, odom, corner_cloud, surf_cloud):
        odom = np.asarray(odom, np.float64).reshape(4, 4)
        if not self.keyframe_updater.update(odom):
            return None
        kf = KeyFrame(np.eye(4), self.keyframe_updater.get_accum_distance(), corner_cloud, surf_cloud,
                      frame_id=self.keyframe_updater.get_unique_id())
        kf.odom = odom.copy()
        kf.node = None
        self.keyframe_queue.append(kf)
        return kf

    # graph.cpp:248-297
    def flush_keyframe_queue(self):
        if not self.keyframe_queue:
            return False
        odom2map = self.tf_odom2graph.copy()
        n = min(len(self.keyframe_queue), self.max_keyframes_per_update)
        for i in range(n):
            kf = self.keyframe_queue[i]
            self.new_keyframes.append(kf)
            kf.estimate = odom2map @ kf.odom
            kf.node = self.solver.add_se3_node(kf.estimate)
            if i == 0 and not self.keyframes:
                continue
            prev = self.keyframes[-1] if i == 0 else self.keyframe_queue[i - 1]
            relative = np.linalg.inv(prev.odom) @ kf.odom
            self.solver.add_se3_edge(prev.node, kf.node, relative, ODOMETRY_INFORMATION)
        del self.keyframe_queue[:n]
        return True

    # one pass of the loop in graph.cpp:313-383
    def optimize(self, max_iterations=1000):
        """-> (loops found in this pass, LM iterations run)."""
        if not self.flush_keyframe_queue():
            return [], 0
        loops = self.loop_detector.detect_nearest(self.keyframes, self.new_keyframes)
        for lp in loops:
            self.solver.add_se3_edge(lp.key1.node, lp.key2.node, lp.relative_pose.astype(np.float64), LOOP_INFORMATION)
        self.loops.extend(loops)
        self.keyframes.extend(self.new_keyframes)
        self.new_keyframes = []
        iterations = 0
        if loops:
            iterations = self.solver.optimize(max_iterations)
            poses = self.solver.poses()
            from .pose_graph import pose7_to_mat
            for kf in self.keyframes:
                kf.estimate = pose7_to_mat(poses[kf.node])
        last = self.keyframes[-1]
        self.tf_odom2graph = last.estimate @ np.linalg.inv(last.odom)
        return loops, iterations
