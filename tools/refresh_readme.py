#!/usr/bin/env python3
"""README.md from tools/readme_template.md, the "Measured" table of DESIGN 4 and the numbered rows of profiles/README.md from the newest kept full report (profiles/rNN_bench_report.json -- what
`python bench.py` wrote to bench_report.json on the GPU box), so that the README's numbers are the kept file's.
    python tools/refresh_readme.py [report.json]"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pick(d, *ks, default=float("nan")):
    for k in ks:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d if d is not None else default


def read_pmc(path):
    """profiles/rNN_pmc_sweep.csv (tools/summarize_pmc.py) -> {counter: mean per launch}"""
    v = {}
    if not os.path.exists(path):
        return v
    for line in open(path):
        f = line.strip().split(",")
        if not line.startswith("#") and len(f) >= 4 and f[0] != "pass":
            v[f[1]] = float(f[3])
    return v


def read_pmc_by_kernel(path):
    """profiles/rNN_pmc_sweep_by_kernel.csv (tools/summarize_pmc_by_name.py) -> {kernel: {counter: mean per launch}}"""
    import csv
    out = {}
    if not os.path.exists(path):
        return out
    rows = [l for l in open(path) if not l.startswith("#")]
    for r in csv.DictReader(rows):
        out.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_launch"])
    return out


def replace_block(path, tag, body):
    """Replace what stands between <!-- GEN:tag --> and <!-- /GEN:tag --> in a file (both markers stay)."""
    text = open(path).read()
    a, b = "<!-- GEN:%s -->" % tag, "<!-- /GEN:%s -->" % tag
    if a not in text or b not in text:
        raise SystemExit("%s has no %s ... %s block" % (path, a, b))
    i, j = text.index(a) + len(a), text.index(b)
    open(path, "w").write(text[:i] + "\n" + body.rstrip() + "\n" + text[j:])


def generated_tables(rnd, d):
    """The numbers DESIGN 4's 'Measured' table and profiles/README.md quote, recomputed from the kept files of round `rnd`:
    round 4's review found three of them drifted from the CSVs beside them."""
    pre = os.path.join(ROOT, "profiles", "r%02d_" % rnd)
    g, lane, byk = read_pmc(pre + "pmc_sweep.csv"), read_pmc(pre + "lane_pmc_sweep.csv"), read_pmc_by_kernel(pre + "pmc_sweep_by_kernel.csv")
    if not g:
        return
    def line(path):
        try:
            return json.loads(open(path).read().strip().splitlines()[-1])
        except Exception:
            return {}
    hl, ll = line(pre + "headline.json"), line(pre + "lane_headline.json")
    ms = lambda v: v.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0 / 2.4e9 * 1e3
    lanes = lambda v: v.get("SQ_THREAD_CYCLES_VALU", float("nan")) / max(1.0, v.get("SQ_INSTS_VALU", 1.0))
    hbm = lambda v: (2.0 * v.get("FETCH_SIZE", float("nan")) + v.get("WRITE_SIZE", float("nan"))) * 1024.0
    k1 = next((v for k, v in byk.items() if k.startswith("sweep_grid_kernel")), {})
    k2 = next((v for k, v in byk.items() if k.startswith("sweep_queue_kernel")), {})
    cs = d.get("certificate_sweep", {})
    gs = d.get("grid_sweep", {})
    pts = pick(hl, "roofline", "points_per_launch")
    rows = [
        "| per sweep (%.1f M points) | kd-tree walk + certificates | grid sweep |" % (pts / 1e6),
        "|---|---|---|",
        "| point-residuals/s, whole step (`r%02d_bench_report.json`) | %.3g (%.3g with every search executed) | **%.3g** |"
        % (rnd, pick(cs, "kd_tree_walk", "value"), pick(cs, "kd_tree_walk", "value_searching_every_point"), d["value"]),
        "| ... of the profiled command (`r%02d_headline.json`, `r%02d_lane_headline.json`: %d steps) | %.3g | %.3g |" % (rnd, rnd, int(hl.get("steps", 0)), ll.get("value", float("nan")), hl.get("value", float("nan"))),
        "| `SQ_INSTS_VALU` | %.3g | **%.3g** (%.3g pass 1 + %.3g pass 2) |" % (lane.get("SQ_INSTS_VALU", float("nan")), g["SQ_INSTS_VALU"], k1.get("SQ_INSTS_VALU", float("nan")), k2.get("SQ_INSTS_VALU", float("nan"))),
        "| lanes active (`SQ_THREAD_CYCLES_VALU` / `SQ_INSTS_VALU`) | %.1f | **%.1f** (%.1f in pass 1, %.1f in pass 2) |" % (lanes(lane), lanes(g), lanes(k1), lanes(k2)),
        "| engine time (`GRBM_GUI_ACTIVE` / 8 / 2.4 GHz) | %.2f ms | **%.2f ms** (%.2f + %.2f) |" % (ms(lane), ms(g), ms(k1), ms(k2)),
        "| HIP events around a sweep (`roofline.avg_kernel_ms` of the same commands) | %.2f ms | %.2f ms |" % (pick(ll, "roofline", "avg_kernel_ms"), pick(hl, "roofline", "avg_kernel_ms")),
        "| points left to the tree search | all (minus the certified ones) | %.1f %% (first sweep of a loop: %.1f %% of the corner points, %.1f %% of the surf points; later sweeps below %.1f %%) |"
        % (100 * pick(gs, "share_left_to_the_tree_search"), 100 * (pick(gs, "share_by_sweep", "corner", default=[float("nan")])[0]),
           100 * (pick(gs, "share_by_sweep", "surf", default=[float("nan")])[0]),
           100 * max([x for x in (pick(gs, "share_by_sweep", "corner", default=[0, 0])[1:5] + pick(gs, "share_by_sweep", "surf", default=[0, 0])[1:5])] or [float("nan")])),
        "| HBM-side traffic ((2 · FETCH + WRITE) · 1 KiB) | — | %.2f GB |" % (hbm(g) / 1e9),
    ]
    replace_block(os.path.join(ROOT, "DESIGN.md"), "grid_table", "\n".join(rows))
    # profiles/README.md: the rows of this round's headline evidence, with the numbers of the files they describe
    chk = ""
    cpath = pre + "profile_check.txt"
    if os.path.exists(cpath):
        chk = " ".join(l.strip() for l in open(cpath) if l.startswith("profiler:"))[:400]
    prow = [
        "| `r%02d_pmc_sweep.csv` | the seven `rocprofv3 --pmc` passes of the headline command, mean **per sweep** (pass 1 + planner + pass 2): `SQ_INSTS_VALU` %.3g, `SQ_THREAD_CYCLES_VALU` / `SQ_INSTS_VALU` = %.1f lanes, `GRBM_GUI_ACTIVE` / 8 / 2.4 GHz = %.2f ms, (2 × FETCH + WRITE) × 1 KiB = %.2f GB; `bench.py` reads `roofline.valu_issue`, `measured_hbm`, `counters` from this file and says so | `tools/collect_profiles.sh r%02d` |"
        % (rnd, g["SQ_INSTS_VALU"], lanes(g), ms(g), hbm(g) / 1e9, rnd),
        "| `r%02d_pmc_sweep_by_kernel.csv` | the first four passes per kernel NAME: `sweep_grid_kernel<256>` (%.1f lanes, %.2f ms), `sweep_queue_kernel` (%.1f lanes, %.2f ms: the sparse subset's tree searches), the planner | same (`tools/summarize_pmc_by_name.py`) |"
        % (rnd, lanes(k1), ms(k1), lanes(k2), ms(k2)),
        "| `r%02d_headline.json`, `r%02d_headline_kernel_stats.csv`, `r%02d_profile_check.txt` | `bench.py --headline-only --steps 6 --warmup 1` (%.3g point-residuals/s, %.3f ms per sweep by HIP events) and `rocprofv3 --kernel-trace --stats` of the same command; the check `collect_profiles.sh` ends with (`tools/check_profile_consistency.py`): %s | same |"
        % (rnd, rnd, rnd, hl.get("value", float("nan")), pick(hl, "roofline", "avg_kernel_ms"), chk or "(not collected)"),
        "| `r%02d_lane_headline.json`, `r%02d_lane_headline_kernel_stats.csv`, `r%02d_lane_pmc_sweep.csv` | the same command with `--search lane` (round 3's kernel + certificate sweep, this round's build): `SQ_INSTS_VALU` %.3g per sweep, %.1f lanes, %.2f ms | same |"
        % (rnd, rnd, rnd, lane.get("SQ_INSTS_VALU", float("nan")), lanes(lane), ms(lane)),
    ]
    replace_block(os.path.join(ROOT, "profiles", "README.md"), "r%02d_numbers" % rnd, "\n".join(prow))
    print("DESIGN.md (grid_table), profiles/README.md (r%02d_numbers) <- profiles/r%02d_*" % (rnd, rnd))


def main():
    rep = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_report.json")))[-1]
    d = json.load(open(rep))
    rnd = int(re.search(r"r(\d\d)_", os.path.basename(rep)).group(1)) if re.search(r"r(\d\d)_", os.path.basename(rep)) else 0
    roof = d.get("roofline", {})
    vals = dict(
        round=rnd, report=os.path.relpath(rep, ROOT), value=d["value"], scans=pick(d, "config", "scans_per_step_per_gpu"),
        map_corner_k=pick(d, "config", "map", "surround_corner") / 1e3, map_surf_k=pick(d, "config", "map", "surround_surf") / 1e3,
        vlp16=pick(d, "vlp16_throughput", "value"), single_ms=pick(d, "single_scan", "ms_per_scanmatch"),
        pose_diff=pick(d, "cpu_baseline", "pose_diff_gpu_vs_cpu_m"), cpu=pick(d, "cpu_baseline", "value"),
        unproven_pct=100.0 * pick(d, "grid_sweep", "share_left_to_the_tree_search"), every=pick(d, "certificate_sweep", "value_searching_every_point"),
        lane_every=pick(d, "certificate_sweep", "kd_tree_walk", "value_searching_every_point"), lane_cert=pick(d, "certificate_sweep", "kd_tree_walk", "value"),
        lanes=pick(roof, "counters", "lanes_active"), valu=pick(roof, "valu_issue", "valu_wave_instructions_per_launch"),
        hbm_pct=100.0 * pick(roof, "measured_hbm", "frac"), mf=pick(d, "mapping_frame", "gpu_ms_per_frame"),
        mf_frames=pick(d, "mapping_frame", "frames", default=0), mf_p99=pick(d, "mapping_frame", "gpu_ms_p99"),
        pg_inexact=pick(d, "pose_graph", "inexact_lm", "lm_iters_per_s"), mf_pipe=pick(d, "mapping_frame", "pipelined", "ms_per_frame"),
        mf16=pick(d, "mapping_frame_vlp16", "gpu_ms_per_frame"), mf_ov=pick(d, "mapping_frame", "overlapped", "gpu_ms_per_frame"),
        mf_cpu=pick(d, "mapping_frame", "cpu_ms_per_frame"), pg=pick(d, "pose_graph", "lm_iters_per_s"), pg_cpu=pick(d, "pose_graph", "cpu_baseline", "value"),
        valu_frac=pick(roof, "valu_issue", "frac"), roof_frac=pick(roof, "frac"),
        chain16=pick(d, "sweep_pipeline", "vlp16", "ms_per_sweep"), chain64=pick(d, "sweep_pipeline", "rings64", "ms_per_sweep"),
        thr16=pick(d, "sweep_pipeline", "vlp16", "node_threads", "ms_per_sweep"), thr64=pick(d, "sweep_pipeline", "rings64", "node_threads", "ms_per_sweep"),
        odo16=pick(d, "sweep_pipeline", "vlp16", "ms", "odometry"), odo64=pick(d, "sweep_pipeline", "rings64", "ms", "odometry"),
        map16=pick(d, "sweep_pipeline", "vlp16", "ms", "mapping"), map64=pick(d, "sweep_pipeline", "rings64", "ms", "mapping"),
        ffm=pick(d, "final_feature_map", "keyframes_per_s"))
    tpl = open(os.path.join(ROOT, "tools", "readme_template.md")).read()
    text = tpl.format(**vals)
    text = re.sub(r"(\d(?:\.\d+)?)e\+?0?(\d+)", r"\1e\2", text)  # 1.27e+10 -> 1.27e10
    open(os.path.join(ROOT, "README.md"), "w").write(text)
    print("README.md <-", os.path.relpath(rep, ROOT))
    generated_tables(rnd, d)


if __name__ == "__main__":
    main()
