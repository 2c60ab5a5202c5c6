#!/usr/bin/env python3
"""The kept rocprofv3 kernel statistics must reproduce the bench line they are kept beside.

usage: check_profile_consistency.py <headline.json> <headline_kernel_stats.csv> [tolerance, default 0.03]

<headline.json>   the line `bench.py --headline-only` printed (roofline.avg_kernel_ms: HIP events around every sweep = the
                  dispatch group sweep_grid_kernel + cert_plan_kernel + sweep_queue_kernel, or sweep_kernel [+ its second pass])
<..._stats.csv>   rocprofv3 --kernel-trace --stats of the SAME command

Checks (exit 1 with a message when one fails -- tools/collect_profiles.sh ends with this):
  1. sum of the sweep kernels' total time / launches of the first kernel, the mean per sweep by the profiler, is within the
     tolerance of roofline.avg_kernel_ms (the profiler's figure leaves out the two launch gaps inside a sweep: it may be
     smaller by up to the tolerance, and not larger by more than it);
  2. that mean x sweep launches per step <= ms_per_step (the contract: dominant kernel time per step within the step).
"""
import csv
import json
import sys

SWEEP_FIRST = ("sweep_grid_kernel", "sweep_kernel")
COMPANIONS = ("sweep_queue_kernel", "cert_plan_kernel")


def profiler_mean_per_sweep(stats_csv):
    rows = list(csv.DictReader(open(stats_csv)))
    first = [r for r in rows if any(("lslam::" + k + "<") in r["Name"] for k in SWEEP_FIRST)]
    if not first:
        raise SystemExit("check_profile_consistency: no sweep kernel in %s" % stats_csv)
    calls = sum(int(r["Calls"]) for r in first)
    total_ns = sum(float(r["TotalDurationNs"]) for r in first)
    total_ns += sum(float(r["TotalDurationNs"]) for r in rows if any(("lslam::" + k) in r["Name"] for k in COMPANIONS))
    return total_ns / calls / 1e6, calls


def main():
    line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
    mean_ms, calls = profiler_mean_per_sweep(sys.argv[2])
    ev_ms = line["roofline"]["avg_kernel_ms"]
    rel = mean_ms / ev_ms - 1.0
    launches_per_step = line["roofline"]["launches_timed"] / line["steps"]
    per_step = mean_ms * launches_per_step
    print("profiler: %.4f ms per sweep over %d sweeps; HIP events of the line: %.4f ms (%+.1f %%); x %.1f sweeps per step = %.2f ms "
          "against ms_per_step %.2f" % (mean_ms, calls, ev_ms, 100 * rel, launches_per_step, per_step, line["ms_per_step"]))
    bad = []
    if abs(rel) > tol:
        bad.append("kernel-stats mean per sweep %.4f ms differs from roofline.avg_kernel_ms %.4f ms by %+.1f %% (tolerance %.0f %%): the "
                   "profiled command does not time the kernel the line reports" % (mean_ms, ev_ms, 100 * rel, 100 * tol))
    if per_step > line["ms_per_step"] * (1.0 + tol):
        bad.append("kernel time per step %.2f ms exceeds ms_per_step %.2f" % (per_step, line["ms_per_step"]))
    if bad:
        print("PROFILE INCONSISTENT:\n  " + "\n  ".join(bad), file=sys.stderr)
        sys.exit(1)
    print("profile consistent with the line")


if __name__ == "__main__":
    main()
