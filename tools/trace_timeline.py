#!/usr/bin/env python3
"""Print the launch timeline of the LAST surround build in a rocprofv3 kernel trace (see bench_treebuild.py):
start offset [us], duration [us], queue, kernel.   python3 tools/trace_timeline.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "fm_gather_kernel" in r["Kernel_Name"]]
start = idx[-2]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    import re
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+|\w+)(?=[<(]|$)", re.sub(r"\(anonymous namespace\)::|lslam::|rocprim::\w+::detail::", "", r["Kernel_Name"]))
    n = (m.group(1) if m else r["Kernel_Name"])[:24]
    print("%8.1f %7.1f q%s %-22s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), n))
