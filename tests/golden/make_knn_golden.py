"""Generates tests/golden/knn_*.npz from the REFERENCE's own nanoflann
(/root/reference/L_SLAM/src/util/nanoflann.hpp, compiled by oracle/Makefile into
oracle/_ref/libref_nanoflann.so).  Run in the authoring container only:

    make -C oracle && python tests/golden/make_knn_golden.py

Each file holds a seeded map (float32 xyz), queries, and the reference's answer
to nearestKSearch(q, 5) (util/nanoflann_pcl.h:150-162): int32 indices and float32
squared distances.  The fixtures are data, not reference source.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import RefNanoflann, have_ref  # noqa: E402


def cases():
    rng = np.random.default_rng(20240601)
    # 1. uniform random cloud, queries inside and outside the bounding box
    pts = rng.uniform(-30, 30, (6000, 3)).astype(np.float32)
    q = rng.uniform(-36, 36, (1500, 3)).astype(np.float32)
    yield "uniform", pts, q
    # 2. voxel-lattice ground plane + wall (exact distance ties, equal coordinates)
    g = np.arange(-12, 12, 0.4, dtype=np.float32)
    X, Y = np.meshgrid(g, g)
    ground = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size, np.float32)], 1)
    Yw, Zw = np.meshgrid(g, np.arange(0, 6, 0.4, dtype=np.float32))
    wall = np.stack([np.full(Yw.size, 5.0, np.float32), Yw.ravel(), Zw.ravel()], 1)
    pts = np.concatenate([ground, wall]).astype(np.float32)
    q = np.concatenate([pts[rng.integers(0, len(pts), 500)] + np.float32(0.2) * (rng.integers(0, 2, (500, 3))).astype(np.float32),
                        rng.uniform(-13, 13, (700, 3)).astype(np.float32) * np.array([1, 1, 0.3], np.float32)])
    yield "lattice_ties", pts, q.astype(np.float32)
    # 3. noisy planar map like a voxel-downsampled surf map, queries near the surface
    n = 8000
    pts = np.stack([rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.normal(0, 0.02, n)], 1).astype(np.float32)
    q = np.stack([rng.uniform(-42, 42, 1500), rng.uniform(-42, 42, 1500), rng.normal(0, 0.3, 1500)], 1).astype(np.float32)
    yield "planar_noisy", pts, q
    # 4. duplicated points (identical coordinates -> ties broken by traversal order)
    base = rng.uniform(-5, 5, (800, 3)).astype(np.float32)
    pts = np.repeat(base, 3, axis=0)
    pts = pts[rng.permutation(len(pts))]
    q = rng.uniform(-6, 6, (800, 3)).astype(np.float32)
    yield "duplicates", pts, q
    # 5. tiny clouds around the leaf size
    for n in (5, 10, 11, 21):
        pts = rng.normal(0, 1, (n, 3)).astype(np.float32)
        q = rng.normal(0, 1.5, (64, 3)).astype(np.float32)
        yield "tiny%d" % n, pts, q


def main():
    if not have_ref():
        raise SystemExit("oracle/_ref/libref_nanoflann.so missing: run `make -C oracle` where /root/reference exists")
    for name, pts, q in cases():
        ref = RefNanoflann(pts)
        idx, d2 = ref.knn(q, 5)
        out = os.path.join(HERE, "knn_%s.npz" % name)
        np.savez_compressed(out, pts=pts, queries=q, idx=idx, d2=d2)
        print(name, pts.shape, q.shape, "->", os.path.basename(out), os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
