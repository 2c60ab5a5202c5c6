#!/usr/bin/env python3
"""Per-launch durations of the sweep kernels in a rocprofv3 kernel trace (csv), in launch order: pass 1 / pass 2 of the
certificate sweep side by side.   python3 tools/sweep_durations.py <trace dir> [last N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sw = [("q" if "sweep_queue" in r["Kernel_Name"] else "s", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
      for r in rows if "sweep_kernel" in r["Kernel_Name"] or "sweep_queue" in r["Kernel_Name"]]
print(len(sw), "sweep launches; last %d [us] (s = sweep_kernel, q = sweep_queue_kernel):" % n, " ".join("%s%.0f" % x for x in sw[-n:]))
