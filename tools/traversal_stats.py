#!/usr/bin/env python3
"""Per-lane traversal statistics of the sweep kernel on the bench workload (one 64-ring scan against the
10k-frame voxel map).  Needs a -DLSLAM_TRAVERSAL_STATS build:

    make -C the-cooper-mapper_amd/csrc EXTRA=-DLSLAM_TRAVERSAL_STATS OBJDIR=../../build/obj_trav OUT=../../build/liblslam_trav.so
    LSLAM_LIB=build/liblslam_trav.so python tools/traversal_stats.py

Prints, for the first (unbounded) sweep of a call and for a later (bounded) one: node / leaf / pop counts per
lane, the same as the maximum over the 64 lanes of a wavefront (what the wavefront pays, lanes run in
lock step), and the lane utilisation  sum(lane work) / (64 x max lane work)  a scheme that refills finished
lanes with new queries could recover."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
import synth_gpu

frames = int(os.environ.get("MAP_FRAMES", "10000"))
world_model = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
lidar = synth_gpu.GpuLidar(world_model, 0)
traj = synth_gpu.loop_trajectory(frames)
ctx = pkg.Context(0)
fm, mapstats = synth_gpu.build_voxel_map(pkg, ctx, lidar, traj, rings=16, progress=None)
fm.update(traj[-1][3:].astype(np.float32))
fm.surround_to_map()
info = ctx.map_info()
print("map: %d corner + %d surf, depth %d" % (info.n_corner, info.n_surf, max(info.depth_corner, info.depth_surf)))
g = traj[-1].copy()
qc, qs = lidar.scan(g, 64, 1800, seed=900000)
ctx.scan_set(qc, qs)
lib = ctx.lib
lib.lslam_debug_sweep_clocks.restype = C.c_int
lib.lslam_debug_sweep_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_uint64), C.c_size_t]
init = synth.perturb_pose(g, seed=99).astype(np.float32)
nb = (len(qc) + 255) // 256 + (len(qs) + 255) // 256
nw = nb * 4
cap = nw + nb * 256 * 2 + 16
names = ["t_desc", "t_leaf", "t_pop", "n_node", "n_leaf", "n_pop", "n_take", "n_popit", "t_take", "n_hit", "n_cand"]


def run(pose, label):
    buf = np.zeros(cap * 4, np.uint64)
    n = lib.lslam_debug_sweep_clocks(ctx.h, pose.ctypes.data_as(C.POINTER(C.c_float)), 0,
                                     buf.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
    assert n == nw, n
    print("== %s" % label)
    stamps = buf[:nw * 4].reshape(nw, 4).astype(np.int64)
    ok = (stamps[:, 3] > stamps[:, 0]) & (stamps[:, 1] > 0) & (stamps[:, 2] > 0)
    ph = np.stack([stamps[ok, 1] - stamps[ok, 0], stamps[ok, 2] - stamps[ok, 1], stamps[ok, 3] - stamps[ok, 2]], 1)
    print("per-wavefront shader-clock stamps: search %.0f, fit + Jacobian %.0f, contraction + block sums %.0f cycles "
          "(shares %.2f / %.2f / %.2f)" % (*ph.mean(0), *(ph.sum(0) / ph.sum())))
    raw = buf[nw * 4: nw * 4 + nb * 256 * 8].reshape(nb * 256, 8)
    valid = (raw[:, 7] >> np.uint64(63)) == 1
    st = np.zeros((nb * 256, 11), np.int64)
    st[:, :6] = raw[:, :6].astype(np.int64)
    st[:, 3] = (raw[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    n_cull = (raw[:, 3] >> np.uint64(32)).astype(np.int64)  # far subtrees skipped by their tight box
    st[:, 4] = (raw[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    st[:, 5] = (raw[:, 5] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    st[:, 9] = (raw[:, 4] >> np.uint64(32)).astype(np.int64)   # leaves with at least one accepted candidate
    st[:, 10] = (raw[:, 5] >> np.uint64(32)).astype(np.int64)  # accepted candidates
    st[:, 6] = (raw[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    st[:, 7] = (raw[:, 6] >> np.uint64(32)).astype(np.int64)
    st[:, 8] = (raw[:, 7] & np.uint64((1 << 63) - 1)).astype(np.int64)
    w = st.reshape(nb * 4, 64, 11)
    wm = w.max(axis=1)
    for i in (3, 4, 9, 10, 5, 6, 7):
        v = st[valid, i]
        util = w[:, :, i].sum() / max(1, 64 * wm[:, i].sum())
        print("%-7s lane mean %7.1f p50 %5.0f p90 %5.0f p99 %5.0f max %5.0f | wave-max mean %7.1f | lane utilisation %.2f" % (
            names[i], v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), wm[:, i].mean(), util))
    print("far subtrees culled by their tight box: lane mean %.2f, wave-max mean %.2f" % (
        n_cull[valid].mean(), n_cull.reshape(nb * 4, 64).max(axis=1).mean()))
    cyc = st[:, 0] + st[:, 1] + st[:, 2] + st[:, 8]
    tot = float(cyc[valid].sum())
    print("share of the search's cycles: descent %.2f, leaves %.2f, pops %.2f, takes %.2f" % tuple(
        st[valid, i].sum() / tot for i in (0, 1, 2, 8)))
    cw = cyc.reshape(nb * 4, 64)
    print("search cycles: lane mean %.0f, wave-max mean %.0f; per node step %.0f, per leaf %.0f, per pop round %.0f" % (
        cyc[valid].mean(), cw.max(axis=1).mean(), wm[:, 0].sum() / max(1, wm[:, 3].sum()),
        wm[:, 1].sum() / max(1, wm[:, 4].sum()), wm[:, 2].sum() / max(1, wm[:, 7].sum())))
    # what a wavefront pays today is ~ sum over its outer rounds of the longest descent + a leaf + the longest pop;
    # with refilled lanes the bound is the lane-sum / 64
    for i, cost in ((3, "node"), (4, "leaf")):
        print("  %s steps: paid by the wavefronts %d, lane-sum/64 %d" % (cost, wm[:, i].sum(), w[:, :, i].sum() // 64))
    return st


def regroup(st, key, label):
    """What the wavefronts would pay if the 256 queries of every workgroup were dealt to its four wavefronts in the order
    of `key` (lightest 64 first): sum over wavefronts of the lane maximum, for node steps, leaf visits and pop rounds."""
    nblk = len(st) // 256
    k = key.reshape(nblk, 256)
    order = np.argsort(k, axis=1, kind="stable")
    out = []
    for i in (3, 4, 7):
        v = np.take_along_axis(st[:, i].reshape(nblk, 256), order, axis=1).reshape(nblk * 4, 64)
        out.append(v.max(axis=1).sum())
    base = [st[:, i].reshape(-1, 64).max(axis=1).sum() for i in (3, 4, 7)]
    print("regrouped by %-32s node steps %.3f  leaf visits %.3f  pop rounds %.3f  of what the wavefronts pay now" % (
        label, out[0] / base[0], out[1] / base[1], out[2] / base[2]))


s1 = run(init, "first sweep of a call (unbounded)")
s2 = run(init, "second sweep, same pose (bounded by the previous neighbours)")
s3 = run(g.astype(np.float32), "sweep at the converged pose (bounded by neighbours found 1 step away)")
mid = (0.7 * g + 0.3 * init).astype(np.float32)
s4 = run(mid, "sweep 30 % of the way back to the initial pose")
work = lambda st: st[:, 4] * 16 + st[:, 3]  # leaf visits, then node steps
regroup(s3, work(s3), "its own work (upper bound)")
regroup(s3, work(s2), "the work of the sweep 0.3 m earlier")
regroup(s4, work(s3), "the work of the previous sweep")
regroup(s4, s3[:, 4], "the previous sweep's leaf visits")
