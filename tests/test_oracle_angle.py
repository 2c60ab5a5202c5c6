"""Pins the oracle's pose-angle state to the reference's own lidar_slam::Angle
(util/Angle.h): the cached std::sin / std::cos of the float radian and `a += x` == Angle(rad + x).
Golden vector from the reference (tests/golden/angle_ref.npz, made by make_angle_golden.py); when
oracle/_ref/libref_angle.so is present the same is checked live on fresh values."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_states(oracle, rad, add):
    import ctypes as C
    lib = oracle.lib
    fp = C.POINTER(C.c_float)
    lib.oracle_pose_sincos.argtypes = [fp, fp]
    lib.oracle_pose_sincos.restype = None
    out = np.zeros((len(rad), 6), np.float32)
    pose = np.zeros(6, np.float32)
    sc = np.zeros(6, np.float32)
    for i, (r, a) in enumerate(zip(rad, add)):
        pose[0] = r
        lib.oracle_pose_sincos(pose.ctypes.data_as(fp), sc.ctypes.data_as(fp))
        out[i, 0], out[i, 1], out[i, 2] = r, sc[0], sc[1]
        s = np.float32(r) + np.float32(a)  # the update of oracle_gn_step: pose[i] = pose[i] + x[i] in fp32
        pose[0] = s
        lib.oracle_pose_sincos(pose.ctypes.data_as(fp), sc.ctypes.data_as(fp))
        out[i, 3], out[i, 4], out[i, 5] = s, sc[0], sc[1]
    return out


def same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or \
        bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_angle_state_matches_reference_golden(oracle):
    g = np.load(os.path.join(ROOT, "tests", "golden", "angle_ref.npz"))
    with np.errstate(over="ignore", invalid="ignore"):
        got = oracle_states(oracle, g["rad"], g["add"])
    assert len(g["rad"]) == 4000
    assert same_bits(got, g["state"])


def test_angle_state_matches_reference_live(oracle):
    ref = os.path.join(ROOT, "oracle", "_ref", "libref_angle.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built (reference absent)")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_angle_golden", os.path.join(ROOT, "tests", "golden", "make_angle_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(7)
    rad = rng.uniform(-4, 4, 3000).astype(np.float32)
    add = rng.normal(0, 0.1, 3000).astype(np.float32)
    assert same_bits(oracle_states(oracle, rad, add), mod.ref_states(rad, add))
