"""Generates tests/golden/angle_ref.npz from the REFERENCE's own lidar_slam::Angle
(/root/reference/L_SLAM/src/util/Angle.h, compiled by oracle/Makefile into oracle/_ref/libref_angle.so).
Run in the authoring container only:

    make -C oracle && python tests/golden/make_angle_golden.py

The file holds seeded float32 (rad, add) pairs and the reference's {rad, sin, cos} of Angle(rad) and of
the same object after `+= add` (Angle.h:17-18,29).  The fixture is data, not reference source.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libref_angle.so")


def ref_states(rad, add):
    lib = C.CDLL(REF)
    lib.ref_angle_state.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float)]
    out = np.zeros((len(rad), 6), np.float32)
    buf = (C.c_float * 6)()
    for i, (r, a) in enumerate(zip(rad, add)):
        lib.ref_angle_state(float(r), float(a), buf)
        out[i] = buf[:]
    return out


def inputs():
    rng = np.random.default_rng(20240602)
    rad = np.concatenate([rng.uniform(-np.pi, np.pi, 3000), rng.normal(0, 0.05, 800), rng.uniform(-50, 50, 180),
                          [0.0, -0.0, np.pi, -np.pi, np.pi / 2, 1e-8, -1e-8, 1e-30, 3.4e38, 7.0, -7.0,
                           0.5 * np.pi, 1.5707964, 100.0, 1e6, 2 ** -126, 1.0, -1.0, 2 * np.pi, 4.0]]).astype(np.float32)
    add = np.concatenate([rng.normal(0, 0.02, 3000), rng.uniform(-0.5, 0.5, 800), rng.uniform(-7, 7, 180),
                          rng.normal(0, 1e-4, 20)]).astype(np.float32)
    return rad, add


if __name__ == "__main__":
    rad, add = inputs()
    out = ref_states(rad, add)
    np.savez_compressed(os.path.join(HERE, "angle_ref.npz"), rad=rad, add=add, state=out)
    print("wrote angle_ref.npz:", out.shape)
