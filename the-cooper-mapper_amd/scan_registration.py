"""Host-side mirror of the feature-extraction front end: ``ScanRegistration::extractFeatures``
(/root/reference/L_SLAM/src/odometry/ScanRegistration.cpp:190-425) over the C ABI
(``lslam_extract_features``, kernels in ``csrc/lslam_features.hip``)."""
import ctypes as C

import numpy as np

from .capi import LslamError, LslamRegParams, c_float_p, c_int32_p

LISTS = ("sharp", "less_sharp", "flat", "less_flat")


class FeatureSet:
    """A sweep's four feature clouds in HBM (``lslam_fset``): what the registration node hands the odometry node
    without a trip through the host.  Filled by :func:`extract_features_dev` or :meth:`upload`; complete when that call
    returns, free for the next fill once the ``DeviceLaserOdometry.process`` that consumed it has returned."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        rc = ctx.lib.lslam_fset_create(ctx.h, C.byref(h))
        if rc < 0:
            raise LslamError(rc, ctx.lib.lslam_last_error().decode())
        self.h = h

    def counts(self):
        c = (C.c_size_t * 4)()
        self.ctx.lib.lslam_fset_counts(self.h, c)
        return dict(zip(LISTS, (int(v) for v in c)))

    def upload(self, sharp, less_sharp, flat, less_flat, ctx=None):
        ctx = ctx or self.ctx
        arrs = [np.ascontiguousarray(a, np.float32).reshape(-1, 4) for a in (sharp, less_sharp, flat, less_flat)]
        args = []
        for a in arrs:
            args += [a.ctypes.data_as(C.c_void_p), len(a)]
        rc = ctx.lib.lslam_fset_upload(ctx.h, self.h, *args, 16)
        if rc < 0:
            raise LslamError(rc, ctx.lib.lslam_last_error().decode())
        return self

    def download(self, which, ctx=None):
        ctx = ctx or self.ctx
        k = LISTS.index(which) if isinstance(which, str) else int(which)
        n = list(self.counts().values())[k]
        out = np.zeros((n, 4), np.float32)
        m = C.c_size_t()
        rc = ctx.lib.lslam_fset_download(ctx.h, self.h, k, out.ctypes.data_as(c_float_p), n, C.byref(m))
        if rc < 0:
            raise LslamError(rc, ctx.lib.lslam_last_error().decode())
        return out

    def close(self):
        if self.h:
            self.ctx.lib.lslam_fset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def extract_features_dev(ctx, cloud, scan_ranges, fset, params=None, intensity_field=3):
    """:func:`extract_features` with the four lists left in HBM (``fset``); returns their sizes."""
    a = np.ascontiguousarray(cloud, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 4:
        raise ValueError("cloud must be (n, >=4) float32")
    r = np.ascontiguousarray(scan_ranges, dtype=np.int32).reshape(-1, 2)
    counts = (C.c_size_t * 4)()
    rc = ctx.lib.lslam_extract_features_dev(ctx.h, a.ctypes.data_as(C.c_void_p), len(a), a.shape[1] * 4, int(intensity_field) * 4,
                                            r.ctypes.data_as(c_int32_p), len(r), C.byref(params) if params is not None else None,
                                            fset.h, counts)
    if rc < 0:
        raise LslamError(rc, ctx.lib.lslam_last_error().decode())
    return dict(zip(LISTS, (int(v) for v in counts)))


def default_params(ctx):
    p = LslamRegParams()
    ctx.lib.lslam_reg_default_params(C.byref(p))
    return p


def extract_features(ctx, cloud, scan_ranges, params=None, intensity_field=3, taps=False):
    """cloud: (n, >=4) float32, xyz first, ``intensity_field`` = column copied to the outputs'
    intensity; scan_ranges: (rings, 2) inclusive [first, last].  Returns a dict with the four feature
    clouds ``sharp``, ``less_sharp``, ``flat``, ``less_flat`` ((m, 4) each) and, with ``taps``, the
    per-point ``curvature``, ``picked`` (marks after setScanBuffersFor) and ``label``."""
    a = np.ascontiguousarray(cloud, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 4:
        raise ValueError("cloud must be (n, >=4) float32")
    r = np.ascontiguousarray(scan_ranges, dtype=np.int32).reshape(-1, 2)
    n = len(a)
    outs = [ctx.scratch("features%d" % k, n, 4) for k in range(4)]
    counts = (C.c_size_t * 4)()
    curv = np.zeros(n, np.float32) if taps else None
    picked = np.zeros(n, np.int8) if taps else None
    label = np.zeros(n, np.int8) if taps else None
    fp = lambda x: x.ctypes.data_as(c_float_p) if x is not None else None
    bp = lambda x: x.ctypes.data_as(C.POINTER(C.c_int8)) if x is not None else None
    rc = ctx.lib.lslam_extract_features(ctx.h, a.ctypes.data_as(C.c_void_p), n, a.shape[1] * 4,
                                        int(intensity_field) * 4, r.ctypes.data_as(c_int32_p), len(r),
                                        C.byref(params) if params is not None else None, fp(outs[0]), fp(outs[1]),
                                        fp(outs[2]), fp(outs[3]), counts, fp(curv), bp(picked), bp(label))
    if rc < 0:
        raise LslamError(rc, ctx.lib.lslam_last_error().decode())
    res = dict(sharp=outs[0][:counts[0]].copy(), less_sharp=outs[1][:counts[1]].copy(),
               flat=outs[2][:counts[2]].copy(), less_flat=outs[3][:counts[3]].copy())
    if taps:
        res.update(curvature=curv, picked=picked, label=label)
    return res


def multiscan_register(ctx, cloud, lower_deg, upper_deg, n_rings, scan_period=0.1):
    """MultiScanRegistration::process (no IMU): raw driver cloud (n, >=3) -> (ring-sorted (m, 4)
    {x', y', z', ring + relTime}, ranges (n_rings, 2)) -- the inputs of :func:`extract_features`."""
    a = np.ascontiguousarray(cloud, dtype=np.float32)
    out = ctx.scratch("multiscan", len(a), 4)
    ranges = np.zeros((int(n_rings), 2), np.int32)
    n = C.c_size_t()
    rc = ctx.lib.lslam_multiscan_register(ctx.h, a.ctypes.data_as(C.c_void_p), len(a), a.shape[1] * 4,
                                          float(lower_deg), float(upper_deg), int(n_rings), float(scan_period),
                                          out.ctypes.data_as(c_float_p), len(a), C.byref(n),
                                          ranges.ctypes.data_as(c_int32_p))
    if rc < 0:
        raise LslamError(rc, ctx.lib.lslam_last_error().decode())
    return out[:n.value].copy(), ranges
