#!/bin/bash
# Per-kernel durations of a few mapping frames (rocprofv3 --kernel-trace --stats over tools/frame_trace.py): run on the GPU box
# after a bench.py run has left its map cache at $1 (default /tmp/mc).
ulimit -c 0
root=$(cd "$(dirname "$0")/.." && pwd)
mc=${1:-/tmp/mc}
( cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ffp -o f -- python3 $root/tools/frame_trace.py --map-cache $mc --frames 60 > /tmp/ffp.log 2>&1 )
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/ffp/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:45]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("lslam::", "").split("(")[0]
        print("%-60s calls %5s  avg %8.1f us  total %9.1f us" % (n[-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
