#!/usr/bin/env python3
"""bench.py -- point-residuals/s of the L_SLAM scan-to-map Gauss-Newton hot path on MI355X.

A "step" is one scanMatchScan Gauss-Newton loop (ScanMatch.cpp:78-347: up to 10
iterations of transform -> kd-tree 5-NN -> line/plane fit -> residual + Jacobian ->
J^T J / J^T r -> 6x6 solve -> pose update) of one synthetic 64-ring x 1800 scan
(115 200 points) against a resident ~1.3 M-point voxel map (BASELINE.json configs[2]);
`--batch` (default 8) different scans are matched together per step, the way keyframes
are re-matched against a map (pose_graph/graph.cpp:171-197).  The single-scan (latency)
figure is reported in the same line under "single_scan".
Map, kd-trees and scans are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]

For N > 1 the driver launches one rank per GPU with torch.distributed.run; the map
is replicated, every rank matches its own scans (independent problems, no data-path
collective: SURVEY.md 8e row 2) and `value` is the whole-job aggregate.

Prints ONE JSON line (rank 0).  `roofline` prices the sweep kernel's algorithmic
bytes (1.7 KB per point-residual) against the 8 TB/s HBM peak, using
the kernel's average duration measured with HIP events on the library's own stream
inside the timed region.  `cpu_baseline` is the oracle (a port of the reference's
single-threaded CPU path, including its per-call kd-tree rebuild) timed on this
host on the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALG_BYTES_PER_POINT_RESIDUAL = 1700.0
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("LSLAM_BATCH", "8")),
                    help="independent scans matched together per step on each GPU")
    ap.add_argument("--jtj-mode", type=int, default=int(os.environ.get("LSLAM_JTJ_MODE", "1")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-joint-stereo", action="store_true")
    ap.add_argument("--no-pose-graph", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the single-scan latency leg")
    ap.add_argument("--no-mapping-frame", action="store_true", help="skip the per-frame mapping pipeline leg")
    ap.add_argument("--shard-points", action="store_true",
                    help="run the sharded-points leg even on one GPU (all-reduce over a world of 1)")
    ap.add_argument("--pg-iters", type=int, default=10, help="LM iterations of the pose-graph leg")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    args = ap.parse_args()

    import numpy as np
    import torch

    distmod = importlib.import_module("the-cooper-mapper_amd.dist")
    rank, local_rank, world = distmod.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    dist = distmod.init("nccl")

    pkg = importlib.import_module("the-cooper-mapper_amd")
    synth = importlib.import_module("the-cooper-mapper_amd.synth")

    # ---- workload: same map on every rank, a different scan pose per rank ----------
    pr = synth.make_problem(rings=args.rings, azimuth_steps=1800, seed=rank)
    ctx = pkg.Context(local_rank)
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    info = ctx.map_info()
    # the batch: `--batch` different scans of the same world (different sensor poses)
    scans, inits, gts = [(pr["corner"], pr["surf"])], [pr["init_pose"]], [pr["gt_pose"]]
    for k in range(1, args.batch):
        gt = (0.01, -0.015, 0.3 + 0.37 * k + 0.1 * rank, 3.0 - 1.7 * k, -2.0 + 2.3 * k, synth.SENSOR_HEIGHT)
        qc, qs, gt = synth.make_scan(pr["world"], args.rings, 1800, gt_pose=gt, seed=1234 + 17 * k + rank)
        scans.append((qc, qs))
        inits.append(synth.perturb_pose(gt, seed=99 + k))
        gts.append(gt.astype(np.float32))
    n_pts = sum(len(c) + len(s) for c, s in scans)
    ctx.scan_set_batch(scans)
    inits = np.stack(inits)
    opts = ctx.default_opts()
    opts.jtj_mode = args.jtj_mode
    opts.profile = 1
    # The interpreter holds ~170 k objects by now (torch, numpy): a generation-2 collection of the Python
    # garbage collector takes 25-35 ms and would land in whichever timed region happens to allocate the
    # triggering object.  Park everything allocated so far in the permanent generation (the same reason
    # timeit disables the collector); the library itself has no Python in its data path.
    quiet_gc()

    def barrier():
        distmod.barrier(dist)

    for _ in range(args.warmup):
        ctx.run_batch(inits, opts)

    barrier()
    t0 = time.perf_counter()
    pt_res = 0
    iters = 0
    sweep_ms = 0.0
    sweep_launches = 0
    loop_ms = 0.0
    last = None
    for _ in range(args.steps):
        status, poses, sts = ctx.run_batch(inits, opts)  # synchronises the library's stream
        pt_res += sum(s.point_residuals for s in sts)
        iters += sum(s.iterations for s in sts)
        sweep_ms += sts[0].gpu_ms_sweep
        sweep_launches += sts[0].sweep_launches
        loop_ms += sts[0].gpu_ms_total
        last = (status, poses, sts)
    barrier()
    elapsed = time.perf_counter() - t0

    (total_pt_res, total_iters), t = distmod.aggregate(dist, [pt_res, iters], elapsed)

    status, poses, sts = last
    pose, st = poses[0], sts[0]
    pose_err = np.abs(poses - np.stack(gts)).max(axis=0)

    if rank == 0:
        avg_sweep_ms = sweep_ms / max(1, sweep_launches)
        # algorithmic bytes of an average timed launch: scans of a batch that have already
        # converged are skipped by later launches, so count the points actually processed
        alg_bytes = ALG_BYTES_PER_POINT_RESIDUAL * pt_res / max(1, sweep_launches)
        achieved_gbs = alg_bytes / (avg_sweep_ms * 1e-3) / 1e9 if avg_sweep_ms > 0 else 0.0
        compulsory = float(info.n_corner + info.n_surf) * 16.0 + (pt_res / max(1, sweep_launches)) * (16.0 + 36.0)
        out = {
            "metric": "point-residuals/s",
            "value": total_pt_res / t,
            "unit": "point-residuals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "lm_iters_per_s": total_iters / t,
            "config": {
                "workload": "synthetic %d-ring x 1800 scan-to-map scanMatchScan GN loop (BASELINE configs[2])" % args.rings,
                "scans_in_flight_per_gpu": args.batch,
                "scan_points": n_pts,
                "scan_corner": int(len(pr["corner"])),
                "scan_surf": int(len(pr["surf"])),
                "map_corner": int(info.n_corner),
                "map_surf": int(info.n_surf),
                "kd_nodes": int(info.nodes_corner + info.nodes_surf),
                "kd_depth": int(max(info.depth_corner, info.depth_surf)),
                "gn_iters_per_step_per_scan": iters / args.steps / args.batch,
                "sweep_launches_per_step": sweep_launches / args.steps,
                "jtj_mode": "mfma_f32_16x16x4" if args.jtj_mode == 1 else "valu_shuffle",
                "parallelism": "replicated map, scans sharded across %d GPU(s), no collective" % world,
                "map_build_ms_outside_timed_region": float(info.build_ms + info.upload_ms),
                "map_built_on_device": bool(info.built_on_device),
                "gpu_loop_ms_per_step": loop_ms / args.steps,
                "pose_err_vs_ground_truth_m": float(pose_err[3:].max()),
                "pose_err_vs_ground_truth_rad": float(pose_err[:3].max()),
                "converged": bool(all(s.converged for s in sts)),
            },
            "roofline": {
                "kernel": "sweep_kernel",
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic_bytes(args.batch),
                "avg_kernel_ms": avg_sweep_ms,
                "launches_timed": sweep_launches,
                "alg_bytes_per_launch": alg_bytes,
                # SURVEY 8d asks for both accountings: the compulsory-unique lower bound of a sweep -- every map
                # point and every query read once, every output written once: (Mc+Ms) 16 B + N (16 + 36) B
                "compulsory_bytes_per_launch": compulsory,
                "compulsory_achieved_gbs": compulsory / (avg_sweep_ms * 1e-3) / 1e9 if avg_sweep_ms > 0 else 0.0,
            },
        }
        if not args.no_single:
            out["single_scan"] = single_scan_leg(ctx, pr, opts, max(10, args.steps // 2))
    if rank == 0 and not args.no_mapping_frame:
        try:
            out["mapping_frame"] = mapping_frame_leg(pkg, synth, ctx, pr, opts, np, not args.no_cpu_baseline, 64)
            # BASELINE configs[1]: the same frame with a VLP-16 (16 x 1800) sweep
            out["mapping_frame_vlp16"] = mapping_frame_leg(pkg, synth, ctx, pr, opts, np, not args.no_cpu_baseline, 16)
        except Exception as e:  # a secondary leg never takes the headline line down
            out["mapping_frame"] = {"error": repr(e)}
    if rank == 0 and not args.no_mapping_frame:
        try:
            out["sweep_pipeline"] = {"vlp16": sweep_pipeline_leg(pkg, synth, ctx, 16, np),
                                     "rings64": sweep_pipeline_leg(pkg, synth, ctx, 64, np)}
        except Exception as e:
            out["sweep_pipeline"] = {"error": repr(e)}
    if world > 1 or args.shard_points:
        try:
            shres = sharded_points_leg(pkg, synth, distmod, dist, rank, world, ctx, opts, torch, np, args)
        except Exception as e:  # never let the secondary leg take the headline line down
            shres = {"error": repr(e)}
        if rank == 0:
            out["sharded_points"] = shres
    if not args.no_joint_stereo:
        try:
            jres = joint_stereo_leg(pkg, synth, distmod, dist, rank, world, ctx, opts, torch, np, args,
                                    not args.no_cpu_baseline)
        except Exception as e:
            jres = {"error": repr(e)}
        if rank == 0:
            out["joint_lidar_stereo"] = jres
    if not args.no_pose_graph:
        pgres = pose_graph_leg(pkg, synth, distmod, dist, rank, world, local_rank, torch, np,
                               args.pg_iters, not args.no_cpu_baseline)
        if rank == 0:
            out["pose_graph"] = pgres
    if rank == 0:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pr, args.cpu_repeats, pose, np)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def quiet_gc():
    import gc
    gc.collect()
    gc.freeze()


def pmc_traffic_bytes(batch):
    """HBM bytes per sweep launch from the committed rocprofv3 --pmc passes of this same
    command (PMC counters cannot be collected inside the timed run; FETCH_SIZE and
    WRITE_SIZE need separate passes).  MI355X_MICROARCH.md: counters are in KiB and
    FETCH_SIZE under-reports wide reads by 2x on gfx950, so traffic = (2*FETCH + WRITE)*1024."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_sweep_batch%d.csv" % batch)
    if not os.path.exists(path):
        return None
    vals = {}
    for line in open(path):
        f = line.strip().split(",")
        if len(f) >= 4 and f[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[f[1]] = float(f[3])
    if len(vals) != 2:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def sharded_points_leg(pkg, synth, distmod, dist, rank, world, ctx, opts, torch, np, args):
    """ONE 115 200-point scan, its points sharded contiguously over the ranks,
    the 32 fp64 normal-equation sums all-reduced over RCCL every Gauss-Newton iteration, the
    6x6 solve replicated.  Latency-bound by construction (two host round trips per iteration);
    reported as measured, next to the recommended no-collective batch mode."""
    pr = synth.make_problem(rings=args.rings, azimuth_steps=1800, seed=0)  # the same scan on every rank
    # ... and the same map: every rank must take the same decisions from the same all-reduced sums
    # (rank 0's map was replaced by the mapping-frame leg)
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    cb, ce = distmod.shard_range(len(pr["corner"]), rank, world)
    sb, se = distmod.shard_range(len(pr["surf"]), rank, world)
    ctx.scan_set(pr["corner"][cb:ce], pr["surf"][sb:se])
    xchg = torch.zeros(32, dtype=torch.float64, device="cuda")

    def allreduce(ptr, count):
        if dist is not None:
            dist.all_reduce(xchg, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    for _ in range(3):
        ctx.run_sharded(pr["init_pose"], allreduce, xchg, opts)
    steps = max(10, args.steps // 2)
    quiet_gc()
    distmod.barrier(dist)
    t0 = time.perf_counter()
    pt = 0
    for _ in range(steps):
        status, pose, st = ctx.run_sharded(pr["init_pose"], allreduce, xchg, opts)
        pt += st.point_residuals
    distmod.barrier(dist)
    dt = time.perf_counter() - t0
    (tot,), tmax = distmod.aggregate(dist, [pt], dt)
    return {"value": tot / tmax, "unit": "point-residuals/s", "ms_per_scanmatch": 1e3 * tmax / steps,
            "gn_iterations": int(st.iterations), "n_gpus": world, "scaling": "strong",
            "allreduce_bytes_per_iteration": 256,
            "pose_err_vs_ground_truth_m": float(np.abs(pose - pr["gt_pose"])[3:].max())}


def joint_stereo_leg(pkg, synth, distmod, dist, rank, world, ctx, opts, torch, np, args, with_cpu):
    """BASELINE configs[4]: LOAM edge/plane rows + stereo reprojection rows in ONE joint 6x6 system per
    Gauss-Newton iteration (include/lslam_c.h lslam_stereo_set; the reference has no code for the visual
    term -- parity unpinned, the oracle restates ORB-SLAM2's pose-only stereo edge).  One 64-ring scan
    and 2 000 stereo observations of map landmarks; at N>1 the scan points AND the observations are
    sharded over the ranks and the 32 sums all-reduced per iteration (the sharded-points path)."""
    pr = synth.make_problem(rings=args.rings, azimuth_steps=1800, seed=0)
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    pts = np.concatenate([pr["map_corner"], pr["map_surf"]])
    lm, ob, w = synth.make_stereo(pts, pr["gt_pose"], n=2000)
    cam = ctx.default_stereo_cam()
    for i, v in enumerate(synth.T_CAM_LIDAR.reshape(-1)):
        cam.T_cl[i] = float(v)
    cam.weight = 1e-3
    cb, ce = distmod.shard_range(len(pr["corner"]), rank, world)
    sb, se = distmod.shard_range(len(pr["surf"]), rank, world)
    o0, o1 = distmod.shard_range(len(lm), rank, world)
    ctx.scan_set(pr["corner"][cb:ce], pr["surf"][sb:se])
    xchg = torch.zeros(32, dtype=torch.float64, device="cuda")

    def allreduce(ptr, count):
        if dist is not None:
            dist.all_reduce(xchg, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()

    def run():
        if world == 1:
            return ctx.run(pr["init_pose"], opts)
        return ctx.run_sharded(pr["init_pose"], allreduce, xchg, opts)

    def timed(steps):
        for _ in range(3):
            run()
        quiet_gc()
        distmod.barrier(dist)
        each = []
        for _ in range(steps):
            t1 = time.perf_counter()
            status, pose, st = run()
            each.append(time.perf_counter() - t1)
        distmod.barrier(dist)
        # median call (max over ranks): one call in a few dozen shows a 30-50 ms host-side stall on the
        # bench box that a mean over 10-25 calls would mostly consist of; the worst call is reported too
        _, med = distmod.aggregate(dist, [0.0], float(np.median(each)))
        _, worst = distmod.aggregate(dist, [0.0], float(max(each)))
        return 1e3 * med, 1e3 * worst, pose, st
    steps = max(10, args.steps // 2)
    ms_lidar, worst_lidar, pose_l, st_l = timed(steps)
    ctx.stereo_set(lm[o0:o1], ob[o0:o1], w[o0:o1], cam)
    ms_joint, worst_joint, pose_j, st_j = timed(steps)
    ctx.stereo_clear()
    n_pts = len(pr["corner"]) + len(pr["surf"])
    res = {"ms_per_joint_scanmatch": ms_joint, "ms_per_lidar_only_scanmatch": ms_lidar,
           "statistic": "median of %d calls (max over ranks)" % steps, "ms_worst_call": [worst_lidar, worst_joint],
           "joint_rows_per_s": (st_j.iterations * (n_pts + 3 * len(lm))) / (1e-3 * ms_joint),
           "scan_points": n_pts, "stereo_observations": int(len(lm)), "stereo_weight": float(cam.weight),
           "gn_iterations": int(st_j.iterations), "gn_iterations_lidar_only": int(st_l.iterations),
           "rows_last_iteration": int(st_j.n_rows), "n_gpus": world,
           "parallelism": "scan points and observations sharded over %d GPU(s), 256 B all-reduce per iteration" % world
           if world > 1 else "one GPU, device-resident loop",
           "pose_err_vs_ground_truth_m": float(np.abs(pose_j - pr["gt_pose"])[3:].max()),
           "pose_err_lidar_only_m": float(np.abs(pose_l - pr["gt_pose"])[3:].max()),
           "parity": "unpinned: no reference code for the visual term"}
    if with_cpu and rank == 0:
        from oracle_lib import Oracle, OracleStereoCam
        o = Oracle(native=True)
        oc = OracleStereoCam()
        for f, _ in OracleStereoCam._fields_:
            setattr(oc, f, getattr(cam, f))
        t0 = time.perf_counter()
        ok, opose, ost, used = o.scanmatch_joint(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                                 lm, ob, w, oc, pr["init_pose"])
        cdt = time.perf_counter() - t0
        res["cpu_baseline"] = {"ms_per_joint_scanmatch": 1e3 * cdt, "cores": 1, "kind": "port",
                               "sample": "one joint call of the oracle (kd-trees rebuilt inside, as the reference's scanMatchScan does)",
                               "gn_iterations": int(ost.iterations), "stereo_observations_used": int(used)}
        if world == 1:
            res["pose_diff_gpu_vs_cpu_m"] = float(np.abs(pose_j - opose)[3:].max())
            res["pose_diff_gpu_vs_cpu_rad"] = float(np.abs(pose_j - opose)[:3].max())
    return res


def pose_graph_leg(pkg, synth, distmod, dist, rank, world, local_rank, torch, np, lm_iters, with_cpu):
    """BASELINE config 4: SE(3) pose-graph LM, 5 000 keyframes / 4 999 odometry + 20 000 loop
    edges, fp64.  Edges are sharded across the ranks; every LM iteration all-reduces the block
    system [H blocks | b | chi2] over RCCL, the damped PCG solve is replicated."""
    g = synth.make_pose_graph()
    pg = pkg.PoseGraph(local_rank)
    pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    ne = len(g["ij"])
    nbytes = 0
    if dist is not None:
        sysbuf = torch.zeros(pg.system_doubles(), dtype=torch.float64, device="cuda")
        base = sysbuf.data_ptr()

        def allreduce(ptr, count):
            off = (ptr - base) // 8
            dist.all_reduce(sysbuf[off:off + count], op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        b, e = distmod.shard_range(ne, rank, world)
        pg.set_shard(b, e, allreduce=allreduce, system_tensor=sysbuf)
        nbytes = sysbuf.numel() * 8
    pg.optimize(1)  # warm-up (also builds the structures); restart from the initial estimate
    pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    if dist is not None:
        pg.set_shard(b, e, allreduce=allreduce, system_tensor=sysbuf)
    t0 = time.perf_counter()
    pg.build()  # graph upload + block structure (g2o: initializeOptimization), outside the timed LM
    build_ms = 1e3 * (time.perf_counter() - t0)
    quiet_gc()
    distmod.barrier(dist)
    t0 = time.perf_counter()
    iters = pg.optimize(lm_iters)
    distmod.barrier(dist)
    dt = time.perf_counter() - t0
    (_,), tmax = distmod.aggregate(dist, [0.0], dt)
    st = pg.last_stats
    res = {"lm_iters_per_s": iters / tmax, "lm_iterations": iters, "lm_trials": st.lm_trials,
           "cg_iterations": st.cg_iterations, "chi2_initial": st.chi2_initial, "chi2_final": st.chi2_final,
           "keyframes": len(g["init"]), "edges": ne, "dtype": "f64", "n_gpus": world,
           "graph_build_ms": build_ms,
           "allreduce_bytes_per_lm_iteration": nbytes,
           "parallelism": "edges sharded over %d GPU(s), RCCL all-reduce of the block system, replicated PCG" % world}
    pg.close()
    if with_cpu and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import posegraph_oracle as po
        t0 = time.perf_counter()
        _, hist = po.optimize(g["init"], g["ij"], g["meas"], g["info"], max_iters=2)
        cdt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": len(hist) / cdt, "unit": "LM-iters/s", "cores": 1, "kind": "port",
                               "sample": "2 LM iterations of the numpy/SuperLU oracle on the same graph "
                                         "(g2o, the reference's solver, is not available here)"}
    return res


def single_scan_leg(ctx, pr, opts, steps):
    """Latency view of the same path: ONE 115 200-point scan in flight (rank 0, after the
    timed region): wall time per scanMatchScan loop and the sweep kernel on its own."""
    ctx.scan_set(pr["corner"], pr["surf"])
    for _ in range(3):
        ctx.run(pr["init_pose"], opts)
    t0 = time.perf_counter()
    pt = 0
    sw_ms = 0.0
    sw_n = 0
    for _ in range(steps):
        status, pose, st = ctx.run(pr["init_pose"], opts)
        pt += st.point_residuals
        sw_ms += st.gpu_ms_sweep
        sw_n += st.sweep_launches
    dt = time.perf_counter() - t0
    avg = sw_ms / max(1, sw_n)
    n = len(pr["corner"]) + len(pr["surf"])
    # PCIe-inclusive: the caller hands over HOST clouds on every call (lslam_scanmatch_scan:
    # pack + Morton order + H2D, then the loop); never the headline value
    t1 = time.perf_counter()
    pt_h = 0
    for _ in range(steps):
        status, pose, st2 = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"], opts)
        pt_h += st2.point_residuals
    dth = time.perf_counter() - t1
    return {"value": pt / dt, "unit": "point-residuals/s", "ms_per_scanmatch": 1e3 * dt / steps,
            "host_buffers_value": pt_h / dth, "host_buffers_ms_per_scanmatch": 1e3 * dth / steps,
            "gn_iterations": st.iterations, "sweep_kernel_ms": avg,
            # algorithmic bytes of the points actually processed / time of ALL sweep launches (the one
            # trailing launch per call that finds the loop finished costs a few microseconds)
            "roofline_frac": ALG_BYTES_PER_POINT_RESIDUAL * pt / (sw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if sw_ms > 0 else None}


def mapping_frame_leg(pkg, synth, ctx, pr, opts, np, with_cpu, rings=64):
    """One LaserMapping frame end to end on the device (SURVEY 8f n1/n2 around the hot path), per step:
    extractFeatures on the 64x1800 sweep, VoxelGrid of the features (LaserMatcher.cpp:289-301, leaf
    1.0 = the reference default), FeatureMap::update, surround -> kd-trees, scanMatchScan,
    addFeatureCloud -- next to the oracle doing the same steps on one host core (one frame)."""
    def xyzi(a):
        o = np.zeros((len(a), 4), np.float32)
        o[:, :3] = a[:, :3]
        return o
    _, _, gt, cloud, ranges = synth.make_scan(pr["world"], rings, 1800, gt_pose=pr["gt_pose"], seed=4321, full=True)
    fm = pkg.FeatureMap(ctx, 21, 11, 21)
    fm.setup_filter_size(0.2, 0.4, 0.6)
    fm.update(gt[3:])
    fm.add_feature_cloud(xyzi(pr["map_corner"]), xyzi(pr["map_surf"]), np.eye(4, dtype=np.float32))
    R, t = synth.pose_to_Rt(gt)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3], T[:3, 3] = R, t
    init = synth.perturb_pose(gt, seed=77, dt=0.1, dr_deg=0.5)
    steps = ("extract_features", "voxel_grid", "update", "surround_to_map", "scan_match", "add_feature_cloud")
    acc = {k: [] for k in steps}
    frames = 6
    quiet_gc()
    for f in range(frames + 1):  # first frame = warm-up
        t0 = time.perf_counter()
        feat = pkg.scan_registration.extract_features(ctx, cloud, ranges)
        t1 = time.perf_counter()
        dc, ds = pkg.voxel_grid(ctx, feat["less_sharp"], 1.0), pkg.voxel_grid(ctx, feat["less_flat"], 1.0)
        t2 = time.perf_counter()
        fm.update(gt[3:])
        t3 = time.perf_counter()
        fm.surround_to_map()
        t4 = time.perf_counter()
        status, pose, st = ctx.scanmatch_scan(dc, ds, init, opts)
        t5 = time.perf_counter()
        fm.add_feature_cloud(dc, ds, T)
        t6 = time.perf_counter()
        if f == 0:
            pose_first = pose.copy()  # same map state as the oracle's single frame below
        if f > 0:
            for k, d in zip(steps, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                acc[k].append(d)
    # median over the frames: a single host-side stall (allocator, scheduler) in one frame would otherwise
    # dominate a 6-frame mean; the worst frame is reported next to it
    gpu = {k: 1e3 * float(np.median(v)) for k, v in acc.items()}
    res = {"rings": rings, "gpu_ms": gpu, "gpu_ms_per_frame": sum(gpu.values()), "frames": frames,
           "gpu_ms_statistic": "median of the timed frames",
           "gpu_ms_worst_frame": 1e3 * float(max(sum(v[i] for v in acc.values()) for i in range(frames))),
           "gpu_ms_per_frame_each": [[round(1e3 * v[i], 3) for v in acc.values()] for i in range(frames)],
           "sweep_points": int(len(cloud)), "features": {k: int(len(v)) for k, v in feat.items()},
           "scan_points_after_voxel_grid": int(len(dc) + len(ds)), "map_points": fm.info()["n_corner"] + fm.info()["n_surf"],
           "scan_match_iterations": int(st.iterations),
           "pose_err_vs_ground_truth_m": float(np.abs(pose[3:] - gt[3:]).max())}
    if with_cpu:
        from oracle_lib import Oracle
        o = Oracle(native=True)
        ofm = o.feature_map(21, 11, 21)
        ofm.setup_filter_size(0.2, 0.4, 0.6)
        ofm.update(gt[3:])
        ofm.add_feature_cloud(xyzi(pr["map_corner"]), xyzi(pr["map_surf"]), np.eye(4, dtype=np.float32))
        t0 = time.perf_counter()
        ofeat = o.extract_features(cloud, ranges)
        t1 = time.perf_counter()
        odc, ods = o.voxel_grid(ofeat["less_sharp"], 1.0), o.voxel_grid(ofeat["less_flat"], 1.0)
        t2 = time.perf_counter()
        ofm.update(gt[3:])
        t3 = time.perf_counter()
        oc, os_ = ofm.get_surround_feature()
        t4 = time.perf_counter()
        ok, opose, ost = o.scanmatch_scan(oc, os_, odc, ods, init)  # kd-trees rebuilt inside (quirk Q4)
        t5 = time.perf_counter()
        ofm.add_feature_cloud(odc, ods, T)
        t6 = time.perf_counter()
        cpu = {k: 1e3 * d for k, d in zip(("extract_features", "voxel_grid", "update", "surround", "scan_match_incl_tree_build",
                                            "add_feature_cloud"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5))}
        res["cpu_ms"] = cpu
        res["cpu_ms_per_frame"] = sum(cpu.values())
        res["cpu"] = {"cores": 1, "kind": "port", "sample": "one frame of the same steps in the oracle"}
        res["pose_diff_gpu_vs_cpu_m"] = float(np.abs(opose[3:] - pose_first[3:]).max())
    fm.close()
    return res


def sweep_pipeline_leg(pkg, synth, ctx, rings, np):
    """The whole per-sweep chain on the device, raw driver cloud in, map pose out:
    MultiScanRegistration::process -> extractFeatures -> LaserOdometry::process ->
    LaserMapping::process (the-cooper-mapper_amd/pipeline.py), milliseconds per stage."""
    lo, hi = (-15.0, 15.0) if rings == 16 else (-24.9, 2.0)
    world = synth.World(half_extent=175.0)
    odo = pkg.LaserOdometry(ctx)
    mapper = pkg.LaserMapping(ctx, cube_dims=(21, 21, 11))
    sr = pkg.scan_registration
    acc = {"register": 0.0, "extract": 0.0, "odometry": 0.0, "mapping": 0.0}
    n = 0
    raws = []
    for k in range(14):  # consecutive sweeps of a moving sensor: 7 for the stage timings, all for the threads
        gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
        _, _, _, cloud, _ = synth.make_scan(world, rings, 1800, gt_pose=gt, seed=300 + k, full=True)
        ring = np.floor(cloud[:, 3]).astype(np.int64)
        raws.append(cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))])  # arrival order of a clockwise sweep
    for k in range(7):
        raw = raws[k]
        quiet_gc()
        t0 = time.perf_counter()
        reg, rr = sr.multiscan_register(ctx, raw, lo, hi, rings)
        t1 = time.perf_counter()
        f = sr.extract_features(ctx, reg, rr)
        t2 = time.perf_counter()
        T = odo.process(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
        t3 = time.perf_counter()
        if T is not None:
            M = mapper.process(odo.last_corner, odo.last_surf, T)
        t4 = time.perf_counter()
        if k >= 2:
            for key, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                acc[key] += d
            n += 1
    mapper.feature_map.close()
    ms = {k: 1e3 * v / n for k, v in acc.items()}
    res = {"rings": rings, "points_per_sweep": int(len(raw)), "ms": ms, "ms_per_sweep": sum(ms.values()),
           "sweeps_per_s": 1e3 / sum(ms.values()), "sweeps_timed": n,
           "travelled_m": float(np.linalg.norm(M[:3, 3]))}
    try:
        res["node_threads"] = sweep_pipeline_threads(pkg, synth, rings, np, lo, hi, raws)
    except Exception as e:
        res["node_threads"] = {"error": repr(e)}
    return res


def sweep_pipeline_threads(pkg, synth, rings, np, lo, hi, raws):
    """The same chain the way the reference runs it: scan registration, odometry and mapping are three
    nodelets with their own threads (nodelets.xml; LaserOdometry.cpp spin(), LaserMapping.cpp:27-37),
    joined by the /laser_cloud_* and /laser_odom_to_init topics.  Three host threads with a context each
    (one call in flight per context), queues between them; ctypes releases the GIL during the calls, the
    three streams share the GPU.  Throughput of the chain, not the latency of a sweep."""
    import queue
    import threading
    sweeps = len(raws)
    ctx_r, ctx_o, ctx_m = pkg.Context(0), pkg.Context(0), pkg.Context(0)
    odo = pkg.LaserOdometry(ctx_o)
    mapper = pkg.LaserMapping(ctx_m, cube_dims=(21, 21, 11))
    sr = pkg.scan_registration
    q1, q2 = queue.Queue(maxsize=2), queue.Queue(maxsize=2)
    out, err = [], []
    warm = 3
    stamps = {}

    def registration():  # MultiScanRegistration nodelet: raw sweep -> feature clouds
        try:
            for k, raw in enumerate(raws):
                if k == warm:
                    stamps["t0"] = time.perf_counter()
                reg, rr = sr.multiscan_register(ctx_r, raw, lo, hi, rings)
                q1.put(sr.extract_features(ctx_r, reg, rr))
        except Exception as e:
            err.append(e)
        q1.put(None)

    def odometry():  # LaserOdometry nodelet
        try:
            while True:
                f = q1.get()
                if f is None:
                    break
                T = odo.process(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
                if T is not None:
                    q2.put((odo.last_corner, odo.last_surf, T))
        except Exception as e:
            err.append(e)
        q2.put(None)

    def mapping():  # LaserMapping nodelet
        try:
            while True:
                item = q2.get()
                if item is None:
                    break
                out.append(mapper.process(*item))
        except Exception as e:
            err.append(e)
        stamps["t1"] = time.perf_counter()
    quiet_gc()
    th = [threading.Thread(target=f) for f in (registration, odometry, mapping)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    mapper.feature_map.close()
    for c in (ctx_r, ctx_o, ctx_m):
        c.close()
    if err:
        raise err[0]
    n = sweeps - warm
    dt = stamps["t1"] - stamps["t0"]
    return {"threads": 3, "ms_per_sweep": 1e3 * dt / n, "sweeps_per_s": n / dt, "sweeps_timed": n,
            "travelled_m": float(np.linalg.norm(out[-1][:3, 3]))}


def cpu_baseline(pr, repeats, gpu_pose, np):
    """The oracle (port of the reference's single-threaded path, -O3 -march=native
    -ffp-contract=off) on the SAME map and scan: full scanMatchScan calls including
    the per-call kd-tree rebuild (quirk Q4)."""
    from oracle_lib import Oracle
    o = Oracle(native=True)
    t0 = time.perf_counter()
    pt = 0
    t_build = t_sweep = 0.0
    its = 0
    for _ in range(repeats):
        ok, pose, st = o.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                        pr["init_pose"])
        pt += st.point_residuals
        t_build += st.t_build
        t_sweep += st.t_sweep
        its += st.iterations
    dt = time.perf_counter() - t0
    # the same loop with the sweep spread over all host cores (OpenMP) -- a generous
    # upper bound for a CPU, NOT what the reference does (its hot path is one thread)
    all_cores = None
    try:
        oo = Oracle(native="omp")
        oo.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"])  # threads up
        t1 = time.perf_counter()
        ok2, pose2, st2 = oo.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"])
        dt2 = time.perf_counter() - t1
        all_cores = {"value": st2.point_residuals / dt2, "sweep_only_value": st2.point_residuals / st2.t_sweep if st2.t_sweep > 0 else None,
                     "cores": os.cpu_count(), "kind": "port, OpenMP over the scan points -- not reference behaviour",
                     "pose_diff_vs_single_thread_m": float(np.abs(pose2[3:] - pose[3:]).max())}
    except Exception as e:
        all_cores = {"error": repr(e)}
    return {
        "all_cores": all_cores,
        "value": pt / dt,
        "unit": "point-residuals/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d full scanMatchScan calls on the same map+scan (%d points, %d GN iterations each), "
                  "kd-tree rebuilt per call as the reference does" % (repeats, len(pr["corner"]) + len(pr["surf"]), its // max(1, repeats)),
        "seconds": dt,
        "sweep_only_value": pt / t_sweep if t_sweep > 0 else None,
        "tree_build_s_per_call": t_build / repeats,
        "host_cpus": os.cpu_count(),
        "pose_diff_gpu_vs_cpu_m": float(np.abs(pose[3:] - gpu_pose[3:]).max()),
        "pose_diff_gpu_vs_cpu_rad": float(np.abs(pose[:3] - gpu_pose[:3]).max()),
        "iterations": its // max(1, repeats),
    }


if __name__ == "__main__":
    main()
