#!/usr/bin/env python3
"""How often a scan point's five neighbours (in order) are the ones of the previous sweep of its Gauss-Newton loop, on the
bench workload: findLine / findPlane are pure functions of the five neighbours in order (util/feature_utils.h:108-204), so a
point whose list did not change needs no new fit.  Per sweep index and feature type: share of points with the same ordered
list, with the same set, and (surf) whose plane passed.  Analysis infrastructure, not product code.

    python tools/nb_change_stats.py [n_scans] [map_frames] [--map-cache PATH]
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_scans = int(argv[0]) if len(argv) > 0 else 6
    frames = int(argv[1]) if len(argv) > 1 else 10000
    cache = None
    if "--map-cache" in sys.argv:
        cache = sys.argv[sys.argv.index("--map-cache") + 1] + ".rank0.npz"
    pkg = importlib.import_module("the-cooper-mapper_amd")
    synth = importlib.import_module("the-cooper-mapper_amd.synth")
    import synth_gpu
    world_model = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
    lidar = synth_gpu.GpuLidar(world_model, 0)
    traj = synth_gpu.loop_trajectory(frames)
    ctx = pkg.Context(0)
    if cache and os.path.exists(cache):
        z = np.load(cache, allow_pickle=True)
        fm = pkg.FeatureMap(ctx, 21, 21, 11)
        fm.setup_filter_size(0.2, 0.4, 0.6)
        fm.update(traj[-1][3:].astype(np.float32))
        fm.add_feature_cloud(z["corner"], z["surf"], np.eye(4, dtype=np.float32))
    else:
        fm, _ = synth_gpu.build_voxel_map(pkg, ctx, lidar, traj, rings=16, progress=2000)
        fm.update(traj[-1][3:].astype(np.float32))
    fm.surround_to_map()
    rng = np.random.default_rng(4242)
    dense = synth_gpu.loop_trajectory(100000)
    seg = np.linalg.norm(np.diff(dense[:, 3:5], axis=0), axis=1).mean()
    span = int(25.0 / seg)
    opts = ctx.default_opts()
    # [type][sweep] counters
    S = 8
    tot = np.zeros((2, S)); same = np.zeros((2, S)); same_set = np.zeros((2, S)); gated = np.zeros((2, S)); fit_ok = np.zeros((2, S))
    for k in range(n_scans):
        g = dense[int(rng.integers(-span, span)) % len(dense)].copy()
        g[3:5] += rng.uniform(-1.0, 1.0, 2)
        g[2] += rng.uniform(-0.2, 0.2)
        qc, qs = lidar.scan(g, 64, 1800, seed=900000 + k)
        init = synth.perturb_pose(g, seed=99 + k)
        ctx.scan_set(qc, qs)
        poses = [np.asarray(init, np.float32)]
        for it in range(1, 11):
            opts.max_iterations = it
            status, pose, st = ctx.run(init, opts)
            if st.iterations < it:
                break
            poses.append(pose.copy())
        # sweep j of the loop runs at poses[j]; the loop ran len(poses) sweeps... the last pose is the result (no sweep there)
        prev = None
        nc = len(qc)
        for j, p in enumerate(poses[:-1] if len(poses) > 1 else poses):
            t = ctx.sweep(p, taps=True, search_mode=1)
            idx, fl = t["idx"], t["flags"]
            for ty, sl in ((0, slice(0, nc)), (1, slice(nc, None))):
                jj = min(j, S - 1)
                i1, f1 = idx[sl], fl[sl]
                tot[ty, jj] += len(i1)
                gated[ty, jj] += int(((f1 & 1) != 0).sum())
                fit_ok[ty, jj] += int(((f1 & 2) != 0).sum())
                if prev is not None:
                    i0, f0 = prev[0][sl], prev[1][sl]
                    both = ((f0 & 1) != 0) & ((f1 & 1) != 0)
                    eq = both & (i0 == i1).all(axis=1)
                    eqs = both & (np.sort(i0, axis=1) == np.sort(i1, axis=1)).all(axis=1)
                    same[ty, jj] += int(eq.sum())
                    same_set[ty, jj] += int(eqs.sum())
            prev = (idx.copy(), fl.copy())
        print("scan %d: %d sweeps" % (k, len(poses) - 1), flush=True)
    for ty, name in ((0, "corner"), (1, "surf")):
        for j in range(S):
            if tot[ty, j] == 0:
                continue
            print("%-6s sweep %d: points %9d  inside the gate %.3f  fit found %.3f  same ordered five as the sweep before %.3f  same set %.3f"
                  % (name, j, tot[ty, j], gated[ty, j] / tot[ty, j], fit_ok[ty, j] / tot[ty, j], same[ty, j] / tot[ty, j], same_set[ty, j] / tot[ty, j]))
    w = tot.sum()
    print("all sweeps >= 1: same ordered five %.3f of the points of those sweeps; of all point-sweeps %.3f"
          % (same[:, 1:].sum() / max(1, tot[:, 1:].sum()), same.sum() / w))


if __name__ == "__main__":
    main()
