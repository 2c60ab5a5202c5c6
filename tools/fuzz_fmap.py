#!/usr/bin/env python3
"""Seeded fuzz of the map maintenance (csrc/lslam_fmap.hip) against the oracle (oracle/fmap_oracle.c), bit for bit:
random cube grids, cube sizes, leaves and active-area radii; clouds of random size with clumps (voxels of hundreds of
members: the centroid kernel's continuation across wavefronts), points outside the grid, repeated points (equal keys);
walks that shift the grid in every direction (cubes entering the active area with unfiltered points: the rebuild's
re-sort fallback) mixed with inserts that do not move (the merge path); the stand-alone VoxelGrid on the same clouds.

    python tools/fuzz_fmap.py [n_seeds]        (N_SEEDS in the environment also works; default 6)

Prints "<n> maps, no mismatch" and exits 0, or the first mismatch and exits 1."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def cloud(rng, n, spread, clumps):
    pts = rng.normal(0, spread, (n, 4)).astype(np.float32)
    pts[:, 2] *= 0.3
    for _ in range(clumps):  # dense clumps: many points per voxel
        k = int(rng.integers(50, 900))
        c = rng.normal(0, spread * 0.5, 3)
        j = rng.integers(0, max(1, n - k))
        pts[j:j + k, :3] = (c + rng.normal(0, float(rng.choice([0.02, 0.1, 0.4])), (min(k, n - j), 3))).astype(np.float32)
    if n > 10 and rng.random() < 0.5:  # exact repeats
        pts[rng.integers(0, n, 5)] = pts[rng.integers(0, n, 5)]
    pts[:, 3] = rng.uniform(0, 16, n)
    return pts


def same(fm, ofm, what):
    a, b = fm.info(), ofm.info()
    if list(a["origin"]) != list(b["origin"]) or list(a["valid"]) != list(b["valid"]):
        return "%s: origin / active area differ" % (what,)
    gc, gs = fm.get_surround_feature()
    oc, os_ = ofm.get_surround_feature()
    if gc.shape != oc.shape or gs.shape != os_.shape or not np.array_equal(bits(gc), bits(oc)) or not np.array_equal(bits(gs), bits(os_)):
        return "%s: surround differs" % (what,)
    return None


def resident_map_is_the_surround(pkg, oracle, ctx, fm, ofm, rng, what):
    """surround_to_map (gathered on the device: lslam_fmap.hip, round 5) against the oracle's surround: the resident map is
    searched at a sample of the surround's own points and at jittered ones -- nanoflann's answer on the oracle's clouds, index
    for index (the indices are positions in the surround)."""
    from oracle_lib import OracleTree
    oc, os_ = ofm.get_surround_feature()
    fm.surround_to_map()
    for which, cloud_ in ((0, oc), (1, os_)):
        if len(cloud_) < 5:
            continue
        pick = rng.integers(0, len(cloud_), min(200, len(cloud_)))
        q = cloud_[pick, :3].copy()
        q[::2] += rng.normal(0, 0.05, q[::2].shape).astype(np.float32)
        try:
            idx, d2 = ctx.knn5(which, q)
        except pkg.LslamError as e:
            return "%s: knn5 on the resident map failed: %s" % (what, e)
        ri, rd = OracleTree(oracle, cloud_).knn(q)
        if not np.array_equal(idx, ri) or not np.array_equal(bits(d2), bits(rd)):
            return "%s: the resident map (type %d, %d points) is not the surround" % (what, which, len(cloud_))
    return None


def one(pkg, oracle, ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    W, H, D = int(rng.integers(5, 14)), int(rng.integers(5, 14)), int(rng.integers(3, 9))
    size = float(rng.choice([6.0, 10.0, 25.0]))
    dist = float(size * rng.uniform(0.9, 2.2))
    leaves = [float(rng.choice([0.1, 0.2, 0.4, 0.8])) for _ in range(3)]
    fm = pkg.FeatureMap(ctx, W, H, D)
    ofm = oracle.feature_map(W, H, D)
    for m in (fm, ofm):
        m.setup_world_cube_size(size)
        m.setup_lidar_valid_distance(dist)
        m.setup_filter_size(*leaves)
    pos = np.zeros(3, np.float32)
    steps = int(rng.integers(5, 10))
    for step in range(steps):
        if step == 0 or rng.random() < 0.55:  # move (sometimes far: shifts, cubes dropping off the rim)
            pos = (pos + rng.normal(0, size * float(rng.choice([0.2, 1.0, 3.0])), 3) * np.array([1, 1, 0.3])).astype(np.float32)
            fm.update(pos)
            ofm.update(pos)
            e = same(fm, ofm, "seed %d step %d after update" % (seed, step))
            if e:
                return e
        for rep in range(int(rng.integers(1, 3))):  # one or two inserts without an update between them
            n = int(rng.integers(0, 6000))
            pts = cloud(rng, n, size * float(rng.uniform(0.3, 1.5)), int(rng.integers(0, 4)))
            nc = int(rng.integers(0, n + 1)) if n else 0
            ang = float(rng.uniform(0, 6.28))
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
            T[:3, 3] = pos
            fm.add_feature_cloud(pts[:nc], pts[nc:], T)
            ofm.add_feature_cloud(pts[:nc], pts[nc:], T)
            e = same(fm, ofm, "seed %d step %d insert %d (%d + %d points)" % (seed, step, rep, nc, n - nc))
            if e:
                return e
            if rng.random() < 0.3:
                e = resident_map_is_the_surround(pkg, oracle, ctx, fm, ofm, rng, "seed %d step %d insert %d" % (seed, step, rep))
                if e:
                    return e
            if n and rng.random() < 0.5:  # the stand-alone filter on the same cloud
                leaf = float(rng.choice([0.2, 0.5, 1.0, 2.0]))
                g = pkg.voxel_grid(ctx, pts, leaf)
                o = oracle.voxel_grid(pts, leaf)
                if g.shape != o.shape or not np.array_equal(bits(g), bits(o)):
                    return "seed %d step %d: VoxelGrid(%g) of %d points differs" % (seed, step, leaf, n)
    if not np.array_equal(bits(fm.get_full_map()), bits(ofm.get_full_map())):
        return "seed %d: full map differs" % seed
    stats = fm.rebuild_stats()
    fm.close()
    return stats


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("N_SEEDS", "6"))
    pkg = importlib.import_module("the-cooper-mapper_amd")
    from oracle_lib import Oracle
    oracle = Oracle()
    ctx = pkg.Context(0)
    ctx.defer_trees(True)  # surround_to_map as the mapping node uses it: cell grids from the gather's own boxes, trees on demand
    merged = resorted = 0
    for seed in range(n):
        r = one(pkg, oracle, ctx, seed)
        if isinstance(r, str):
            print("MISMATCH", r)
            return 1
        merged += r[0]
        resorted += r[1]
    ctx.close()
    print("%d maps, no mismatch (rebuilds: %d merged, %d re-sorted)" % (n, merged, resorted))
    return 0


if __name__ == "__main__":
    sys.exit(main())
