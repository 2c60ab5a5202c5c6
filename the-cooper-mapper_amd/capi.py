"""ctypes binding of include/lslam_c.h (one declaration per exported symbol)."""
import ctypes as C
import enum
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "liblslam_hip.so"

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class LslamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lslam status %d: %s" % (code, msg))
        self.code = code


class Status(enum.IntEnum):
    OK = 0
    TOO_FEW_REF = 1
    NOT_CONVERGED = 2
    LOW_SCORE = 3
    LOW_PERCENT = 4
    TOO_FEW_MATCHES = 5
    ERR_INVALID = -1
    ERR_HIP = -2
    ERR_NO_MAP = -3
    ERR_NO_SCAN = -4
    ERR_TREE_DEPTH = -5
    ERR_TREE_BUILD = -6
    ERR_COMM = -7


class LslamOpts(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32),
        ("delta_t_abort", C.c_float),
        ("delta_r_abort", C.c_float),
        ("use_score", C.c_int32),
        ("fine_score", C.c_int32),
        ("score_threshold", C.c_double),
        ("match_percentage_threshold", C.c_double),
        ("jtj_mode", C.c_int32),
        ("profile", C.c_int32),
        ("scans_in_flight", C.c_int32),
        ("search_mode", C.c_int32),
        ("knn_cert", C.c_int32),
        ("cert_try_m", C.c_float),
        ("cert_track_m", C.c_float),
        ("grid_cell", C.c_float),
        ("debug_stats", C.c_int32),
        ("ab_switches", C.c_int32),
    ]


class LslamRegParams(C.Structure):
    """lslam_reg_params (include/lslam_c.h)."""
    _fields_ = [("n_feature_regions", C.c_int32), ("curvature_region", C.c_int32), ("max_corner_sharp", C.c_int32),
                ("max_surface_flat", C.c_int32), ("less_flat_filter_size", C.c_float),
                ("surface_curvature_threshold", C.c_float), ("blind_threshold", C.c_float), ("reserved", C.c_int32)]


class LslamStereoCam(C.Structure):
    """lslam_stereo_cam (include/lslam_c.h)."""
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("T_cl", C.c_float * 12), ("weight", C.c_float), ("huber_stereo", C.c_float),
                ("huber_mono", C.c_float), ("gate_outliers", C.c_int32), ("min_depth", C.c_float)]


class LslamStats(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("iterations", C.c_int32),
        ("n_line", C.c_int32),
        ("n_plane", C.c_int32),
        ("n_rows", C.c_int32),
        ("degenerate", C.c_int32),
        ("converged", C.c_int32),
        ("delta_r", C.c_float),
        ("delta_t", C.c_float),
        ("score", C.c_double),
        ("percent", C.c_double),
        ("point_residuals", C.c_int64),
        ("sweeps", C.c_int32),
        ("sweep_launches", C.c_int32),
        ("gpu_ms_total", C.c_float),
        ("gpu_ms_sweep", C.c_float),
        ("score2", C.c_double),
        ("percent2", C.c_double),
    ]


class LslamMapInfo(C.Structure):
    _fields_ = [
        ("n_corner", C.c_uint64),
        ("n_surf", C.c_uint64),
        ("nodes_corner", C.c_uint32),
        ("nodes_surf", C.c_uint32),
        ("depth_corner", C.c_int32),
        ("depth_surf", C.c_int32),
        ("build_ms", C.c_float),
        ("upload_ms", C.c_float),
        ("built_on_device", C.c_int32),
        ("build_attempts", C.c_int32),
    ]


class LslamPgStats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("lm_trials", C.c_int32),
        ("cg_iterations", C.c_int32),
        ("status", C.c_int32),
        ("chi2_initial", C.c_double),
        ("chi2_final", C.c_double),
        ("lambda_", C.c_double),
        ("gpu_ms_total", C.c_float),
        ("fused_solves", C.c_int32),
    ]


class LslamOdomStats(C.Structure):
    """lslam_odom_stats (include/lslam_c.h)."""
    _fields_ = [("matched", C.c_int32), ("tree_fallbacks", C.c_int32), ("searches", C.c_int32), ("reserved", C.c_int32),
                ("sweeps", C.c_uint64), ("n_last_corner", C.c_size_t), ("n_last_surf", C.c_size_t)]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t)
ALLGATHERV_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32)
c_double_p = C.POINTER(C.c_double)

# every symbol include/lslam_c.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "lslam_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "lslam_ctx_destroy": (None, [C.c_void_p]),
    "lslam_last_error": (C.c_char_p, []),
    "lslam_default_opts": (None, [C.POINTER(LslamOpts)]),
    "lslam_abi_version": (C.c_int, []),
    "lslam_sizeof_opts": (C.c_size_t, []),
    "lslam_sizeof_stats": (C.c_size_t, []),
    "lslam_map_set": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t]),
    "lslam_map_epoch": (C.c_uint64, [C.c_void_p]),
    "lslam_map_info_get": (C.c_int, [C.c_void_p, C.POINTER(LslamMapInfo)]),
    "lslam_cubemap_set": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                    C.c_float, c_int32_p, c_int32_p]),
    "lslam_scan_set": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t]),
    "lslam_scan_set_batch": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t]),
    "lslam_scanmatch_run_batch": (C.c_int, [C.c_void_p, C.c_int32, c_float_p, C.POINTER(LslamOpts),
                                            C.POINTER(LslamStats)]),
    "lslam_scanmatch_run": (C.c_int, [C.c_void_p, c_float_p, C.POINTER(LslamOpts), C.POINTER(LslamStats)]),
    "lslam_scanmatch_run_sharded": (C.c_int, [C.c_void_p, c_float_p, C.POINTER(LslamOpts), ALLREDUCE_FN,
                                              C.c_void_p, C.c_void_p, C.POINTER(LslamStats)]),
    "lslam_stereo_default_cam": (None, [C.POINTER(LslamStereoCam)]),
    "lslam_stereo_set": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p, C.c_size_t, C.POINTER(LslamStereoCam)]),
    "lslam_stereo_clear": (C.c_int, [C.c_void_p]),
    "lslam_stereo_sums": (C.c_int, [C.c_void_p, c_float_p, c_double_p]),
    "lslam_scanmatch_scan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_size_t, c_float_p, C.POINTER(LslamOpts), C.POINTER(LslamStats)]),
    "lslam_scanmatch_full": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                       c_float_p, C.POINTER(LslamOpts), C.POINTER(LslamStats)]),
    "lslam_odometry_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, c_float_p, C.c_int32,
                                       C.c_float, C.c_float, C.POINTER(LslamStats)]),
    "lslam_odometry_match_trees": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, c_float_p, C.c_int32,
                                             C.c_float, C.c_float, C.POINTER(LslamStats)]),
    "lslam_transform_to_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, c_float_p]),
    "lslam_isometry_to_pose": (None, [c_float_p, c_float_p]),
    "lslam_pose_to_isometry": (None, [c_float_p, c_float_p]),
    "lslam_transform_associate": (None, [c_float_p, c_float_p, c_float_p, c_float_p]),
    "lslam_knn5": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, c_int32_p, c_float_p]),
    "lslam_sweep": (C.c_int, [C.c_void_p, c_float_p, C.c_int32, c_int32_p, c_float_p, c_float_p,
                              c_uint8_p, c_float_p]),
    "lslam_knn5_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int32, c_int32_p, c_float_p,
                                c_int32_p]),
    "lslam_sweep_ex": (C.c_int, [C.c_void_p, c_float_p, C.c_int32, C.c_int32, c_int32_p, c_float_p, c_float_p,
                                 c_uint8_p, c_float_p]),
    "lslam_residuals": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_uint8_p, c_float_p]),
    "lslam_scanmatch_batch": (C.c_int, [C.c_void_p, C.c_int32, c_float_p, C.POINTER(LslamOpts), C.POINTER(LslamStats)]),
    "lslam_posegraph_optimize": (C.c_int, [C.c_int, C.c_int32, c_double_p, C.c_int32, c_int32_p, c_double_p, c_double_p,
                                           C.c_int32, C.c_int32, C.POINTER(LslamPgStats)]),
    "lslam_gn_step": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int32, c_float_p, c_float_p,
                                c_int32_p, C.c_float, C.c_float, c_float_p, c_float_p, c_float_p,
                                c_int32_p]),
    "lslam_stream": (C.c_void_p, [C.c_void_p]),
    "lslam_debug_sweep_launches": (None, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "lslam_debug_cert_stats": (None, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "lslam_debug_grid_stats": (None, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "lslam_debug_knn5_wide": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int32, c_int32_p, c_float_p,
                                        C.POINTER(C.c_uint8)]),
    "lslam_debug_sort_pairs": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint32)]),
    "lslam_debug_cert_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t]),
    "lslam_pg_create": (C.c_int, [C.c_int, C.c_int32, c_double_p, C.c_int32, c_int32_p, c_double_p, c_double_p,
                                  C.c_int32, C.POINTER(C.c_void_p)]),
    "lslam_pg_destroy": (None, [C.c_void_p]),
    "lslam_pg_last_error": (C.c_char_p, []),
    "lslam_fmap_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "lslam_fmap_destroy": (None, [C.c_void_p]),
    "lslam_fmap_setup_filter_size": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "lslam_fmap_setup_world_origin": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "lslam_fmap_setup_world_cube_size": (C.c_int, [C.c_void_p, C.c_float]),
    "lslam_fmap_setup_lidar_valid_distance": (C.c_int, [C.c_void_p, C.c_float]),
    "lslam_fmap_update": (C.c_int, [C.c_void_p, c_float_p]),
    "lslam_fmap_add_feature_cloud": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_size_t, c_float_p]),
    "lslam_fmap_add_feature_cloud_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_size_t, c_float_p]),
    "lslam_fmap_wait": (C.c_int, [C.c_void_p]),
    "lslam_fmap_surround_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lslam_fmap_get_surround": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t, c_float_p, C.c_size_t]),
    "lslam_fmap_surround_to_map": (C.c_int, [C.c_void_p]),
    "lslam_fmap_surround_to_map_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lslam_fmap_to_cubemap": (C.c_int, [C.c_void_p]),
    "lslam_fmap_cubemap_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "lslam_fmap_rebuild_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "lslam_fmap_cubemap_invalidate": (C.c_int, [C.c_void_p]),
    "lslam_fmap_get_full_map": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lslam_fmap_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "lslam_fmap_load": (C.c_int, [C.c_void_p, C.c_char_p]),
    "lslam_fmap_info": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, c_int32_p, C.c_size_t,
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lslam_voxel_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, c_float_p,
                                   C.c_size_t, C.POINTER(C.c_size_t)]),
    "lslam_voxel_grid2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, c_float_p,
                                    C.c_size_t, C.POINTER(C.c_size_t), c_float_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lslam_reg_default_params": (None, [C.POINTER(LslamRegParams)]),
    "lslam_extract_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, c_int32_p,
                                         C.c_size_t, C.POINTER(LslamRegParams), c_float_p, c_float_p, c_float_p,
                                         c_float_p, C.POINTER(C.c_size_t), c_float_p, C.POINTER(C.c_int8),
                                         C.POINTER(C.c_int8)]),
    "lslam_multiscan_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_float,
                                           C.c_int32, C.c_float, c_float_p, C.c_size_t, C.POINTER(C.c_size_t),
                                           c_int32_p]),
    "lslam_fset_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "lslam_fset_destroy": (None, [C.c_void_p]),
    "lslam_fset_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "lslam_fset_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.c_size_t]),
    "lslam_fset_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, c_float_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lslam_extract_features_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, c_int32_p, C.c_size_t,
                                             C.POINTER(LslamRegParams), C.c_void_p, C.POINTER(C.c_size_t)]),
    "lslam_odom_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.POINTER(C.c_void_p)]),
    "lslam_odom_destroy": (None, [C.c_void_p]),
    "lslam_odom_process": (C.c_int, [C.c_void_p, C.c_void_p, c_float_p, c_float_p, C.POINTER(LslamStats), C.POINTER(LslamOdomStats),
                                     c_float_p, C.c_size_t, c_float_p, C.c_size_t]),
    "lslam_odom_last_clouds": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t, c_float_p, C.c_size_t]),
    "lslam_odom_reset": (C.c_int, [C.c_void_p]),
    "lslam_debug_odom_search": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t]),
    "lslam_odom_set_publish": (C.c_int, [C.c_void_p, C.c_int32]),
    "lslam_odom_last_view": (C.c_int, [C.c_void_p, C.POINTER(c_float_p), C.POINTER(C.c_size_t), C.POINTER(c_float_p),
                                       C.POINTER(C.c_size_t)]),
    "lslam_pg_save_g2o": (C.c_int, [C.c_void_p, C.c_char_p]),
    "lslam_g2o_read": (C.c_int, [C.c_char_p, c_int32_p, c_double_p, c_int32_p, c_int32_p, c_double_p, c_double_p,
                                 c_int32_p]),
    "lslam_pg_set_shard": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, ALLREDUCE_FN, C.c_void_p, C.c_void_p]),
    "lslam_pg_system_doubles": (C.c_size_t, [C.c_void_p]),
    "lslam_pg_row_shard_range": (None, [C.c_int32, C.c_int32, C.c_int32, c_int32_p, c_int32_p]),
    "lslam_pg_set_row_shard": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "lslam_pg_row_sharded_solves": (C.c_int32, [C.c_void_p]),
    "lslam_pg_set_solve_tolerance": (C.c_int, [C.c_void_p, C.c_double]),
    "lslam_pg_row_gathered_solves": (C.c_int32, [C.c_void_p]),
    "lslam_pg_set_row_gather": (C.c_int, [C.c_void_p, ALLGATHERV_FN, C.c_void_p, C.c_int32, C.c_int32]),
    "lslam_pg_num_offdiag": (C.c_int32, [C.c_void_p]),
    "lslam_pg_optimize": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(LslamPgStats)]),
    "lslam_pg_get_poses": (C.c_int, [C.c_void_p, c_double_p]),
    "lslam_pg_linearize": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_int32_p, c_double_p, c_double_p]),
    "lslam_pg_solve": (C.c_int, [C.c_void_p, C.c_double, c_double_p, c_int32_p]),
    "lslam_icp_align": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, c_float_p, C.c_int32,
                                  C.c_double, C.c_double, c_double_p, c_int32_p, c_int32_p]),
    "lslam_comm_unique_id": (C.c_int, [c_uint8_p]),
    "lslam_comm_create": (C.c_int, [C.c_int, c_uint8_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "lslam_comm_destroy": (None, [C.c_void_p]),
    "lslam_comm_version": (C.c_int, [c_int32_p]),
    "lslam_comm_info": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p]),
    "lslam_debug_grid_launches": (C.c_uint64, [C.c_void_p]),
    "lslam_debug_grid_wide_launches": (C.c_uint64, [C.c_void_p]),
    "lslam_debug_grid_cells": (None, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "lslam_map_defer_trees": (C.c_int, [C.c_void_p, C.c_int32]),
    "lslam_debug_lazy_trees": (None, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "lslam_comm_allreduce_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lslam_ctx_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lslam_pg_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
}

COMM_ID_BYTES = 128
SEARCH_AUTO, SEARCH_LANE, SEARCH_PACKET, SEARCH_GRID = 0, 1, 2, 3
AB_PERSISTENT_GN, AB_FUSED_SOLVE, AB_SECOND_PROBE, AB_WIDE_IN_PLACE = 1, 2, 4, 8  # lslam_opts.ab_switches (LSLAM_AB_*)
AB_REFILL = 128
STACK_AUTO, STACK_DEEP, STACK_SHALLOW = 0, 0x100, 0x200  # ORed into a search mode (LSLAM_STACK_*)
# lslam_debug_sweep_launches: index of each sweep-kernel instantiation
SWEEP_VARIANTS = ("deep", "deep_ovf", "shallow", "cubes", "cubes_ovf", "packet", "persistent", "deep_fused")


def lib_path():
    # LSLAM_LIB lets profiling scripts load an experimental build of the same ABI
    return os.environ.get("LSLAM_LIB") or os.path.join(HERE, LIB_NAME)


def build_library(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC], stdout=subprocess.DEVNULL)
    return lib_path()


_lib = None


def _preload_torch_runtime():
    """The PyTorch-ROCm wheel ships its own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's).
    A process gets whichever copy is loaded first: this library runs on either, torch only on its own
    ("No HIP GPUs are available" otherwise).  So when torch is installed and not loaded yet, load it
    before liblslam_hip.so -- harnesses (tests, bench.py) use both in one process.  A C/C++ host that
    links liblslam_hip.so never sees this.  LSLAM_NO_TORCH_PRELOAD=1 skips it."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("LSLAM_NO_TORCH_PRELOAD"):
        return
    try:
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:  # a broken torch install must not take the backend down
        pass


def load_library():
    """dlopen liblslam_hip.so and bind every declared symbol.  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build it with `make -C %s` (hipcc --offload-arch=gfx950). "
            "This backend has no CPU fallback." % (path, CSRC))
    _preload_torch_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.lslam_abi_version() < 0:
        raise ImportError("%s was built with LSLAM_EXP_* timing-experiment macros (wrong results on purpose): not a product library; "
                          "LSLAM_ALLOW_EXPERIMENT_BUILD=1 lets the A/B scripts load it" % path)
    # the ctypes mirrors of the ABI's structs must be the library's: a stale .so (or a stale mirror) fails here, loudly
    if lib.lslam_sizeof_opts() != C.sizeof(LslamOpts) or lib.lslam_sizeof_stats() != C.sizeof(LslamStats):
        raise ImportError("%s does not match this package's struct layouts (lslam_opts %d vs %d bytes, lslam_stats %d vs %d): rebuild it"
                          % (path, lib.lslam_sizeof_opts(), C.sizeof(LslamOpts), lib.lslam_sizeof_stats(), C.sizeof(LslamStats)))
    _lib = lib
    return lib
