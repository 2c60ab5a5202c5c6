"""ctypes wrapper of oracle/posegraph_oracle.c (TEST INFRASTRUCTURE ONLY): the compiled pose-graph LM with analytic
Jacobians and a sparse block Cholesky -- the CPU baseline of bench.py's pose-graph leg, checked against
posegraph_oracle.py in tests/test_oracle_posegraph_c.py."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class PgoStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("trials", C.c_int), ("status", C.c_int), ("bandwidth", C.c_int),
                ("env_blocks", C.c_longlong), ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_", C.c_double),
                ("t_linearize", C.c_double), ("t_factor", C.c_double), ("t_solve", C.c_double), ("t_chi2", C.c_double),
                ("t_order", C.c_double), ("t_total", C.c_double), ("factor_flops", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libposegraph_oracle.so")
        src = os.path.join(HERE, "posegraph_oracle.c")
        if os.environ.get("LSLAM_ORACLE_SANITIZE") == "1":  # the ASan + UBSan build (make -C oracle asan)
            path = os.path.join(HERE, "_asan", "libposegraph_oracle.so")
            src = path
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", HERE, "libposegraph_oracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.pgo_chi2.restype = C.c_double
        L.pgo_chi2.argtypes = [C.c_int, dp, ip, dp, dp]
        L.pgo_linearize.argtypes = [C.c_int, dp, C.c_int, ip, dp, dp, C.c_int, dp, dp, dp]
        L.pgo_solve.argtypes = [C.c_int, dp, C.c_int, ip, dp, dp, C.c_int, C.c_double, dp]
        L.pgo_optimize.argtypes = [C.c_int, dp, C.c_int, ip, dp, dp, C.c_int, C.c_int, C.POINTER(PgoStats)]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _args(poses, ij, meas, info):
    poses = np.ascontiguousarray(poses, np.float64)
    ij = np.ascontiguousarray(ij, np.int32)
    meas = np.ascontiguousarray(meas, np.float64)
    info = np.ascontiguousarray(info, np.float64).reshape(len(ij), 36)
    return poses, ij, meas, info


def linearize(poses, ij, meas, info, fixed=0):
    """-> diagonal blocks (n_v, 6, 6), b (6 n_v), chi2"""
    poses, ij, meas, info = _args(poses, ij, meas, info)
    n_v = len(poses)
    diag = np.zeros((n_v, 6, 6))
    b = np.zeros(6 * n_v)
    c2 = C.c_double()
    lib().pgo_linearize(n_v, _d(poses), len(ij), ij.ctypes.data_as(C.POINTER(C.c_int32)), _d(meas), _d(info), fixed, _d(diag), _d(b),
                        C.byref(c2))
    return diag, b, c2.value


def solve(poses, ij, meas, info, lam, fixed=0):
    poses, ij, meas, info = _args(poses, ij, meas, info)
    dx = np.zeros(6 * len(poses))
    rc = lib().pgo_solve(len(poses), _d(poses), len(ij), ij.ctypes.data_as(C.POINTER(C.c_int32)), _d(meas), _d(info), fixed,
                         float(lam), _d(dx))
    if rc:
        raise RuntimeError("pgo_solve: factorisation failed")
    return dx


def optimize(poses, ij, meas, info, fixed=0, max_iters=50):
    """-> (poses, PgoStats)"""
    poses, ij, meas, info = _args(poses, ij, meas, info)
    out = poses.copy()
    st = PgoStats()
    lib().pgo_optimize(len(out), _d(out), len(ij), ij.ctypes.data_as(C.POINTER(C.c_int32)), _d(meas), _d(info), fixed, int(max_iters),
                       C.byref(st))
    return out, st
