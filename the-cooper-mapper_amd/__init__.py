"""MI355X-native nonlinear-least-squares backend for L_SLAM (scan-match hot path).

Python host-side mirror of the reference's ``lidar_slam::ScanMatch``
(/root/reference/L_SLAM/src/scan_to_scan_match/ScanMatch.h:21-61) on top of the
C ABI in ``include/lslam_c.h``.  The arithmetic runs in hand-written HIP kernels
(``csrc/``, built into ``liblslam_hip.so`` next to this file).  There is no CPU
fallback: if the library is missing, importing :mod:`.scan_match` raises, and if
no HIP device is present every compute call raises :class:`LslamError`.

The directory name contains a hyphen, so import it with
``importlib.import_module("the-cooper-mapper_amd")`` (tests/conftest.py and
``__graft_entry__`` register it as ``cooper_mapper_amd`` in ``sys.modules``).
"""
from .capi import (LslamError, LslamOpts, LslamStats, LslamMapInfo, LslamStereoCam, Status, lib_path, load_library,
                   build_library)
from .scan_match import Comm, Context, ScanMatch
from .pose_graph import PoseGraph
from .feature_map import FeatureMap, voxel_grid, voxel_grid2
from . import scan_registration
from .loop_closure import KeyFrame, Loop, LoopDetector
from .graph import Graph, KeyframeUpdater
from .pipeline import DeviceLaserOdometry, LaserOdometry, LaserMapping

__all__ = ["Comm", "Context", "ScanMatch", "PoseGraph", "FeatureMap", "voxel_grid", "voxel_grid2", "scan_registration", "KeyFrame", "Loop", "LoopDetector", "Graph", "KeyframeUpdater", "LaserOdometry", "DeviceLaserOdometry", "LaserMapping", "LslamError", "LslamOpts", "LslamStats", "LslamMapInfo", "LslamStereoCam",
           "Status", "lib_path", "load_library", "build_library"]
