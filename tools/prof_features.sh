#!/bin/bash
# Kernel durations of lslam_extract_features (rocprofv3 --kernel-trace --stats over tools/bench_features.py): run on the GPU box.
ulimit -c 0
root=$(cd "$(dirname "$0")/.." && pwd)
( cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fxp -o fx -- python3 $root/tools/bench_features.py > /tmp/fxp.log 2>&1 )
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/fxp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("fx_", "Buffer", "fm_", "rocprim")):
            print("%-50s calls %4s  avg %8.1f us  min %8.1f  max %8.1f" % (n.replace("(anonymous namespace)::", "").split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
