#!/usr/bin/env python3
"""README.md from tools/readme_template.md and the newest kept full report (profiles/rNN_bench_report.json -- what
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
        mf_cpu=pick(d, "mapping_frame", "cpu_ms_per_frame"), pg=pick(d, "pose_graph", "lm_iters_per_s"), pg_cpu=pick(d, "pose_graph", "cpu_baseline", "value"))
    tpl = open(os.path.join(ROOT, "tools", "readme_template.md")).read()
    text = tpl.format(**vals)
    text = re.sub(r"(\d(?:\.\d+)?)e\+?0?(\d+)", r"\1e\2", text)  # 1.27e+10 -> 1.27e10
    open(os.path.join(ROOT, "README.md"), "w").write(text)
    print("README.md <-", os.path.relpath(rep, ROOT))


if __name__ == "__main__":
    main()
