#!/bin/bash
# quick A/B of the headline + single-scan legs on the GPU box: tools/ab_bench.sh <tag> [env assignments...]
tag=$1; shift
F="--steps 5 --warmup 2 --no-cpu-baseline --no-mapping-frame --no-pipeline --no-joint-stereo --no-pose-graph --map-frames ${MAPF:-3000} ${EXTRA_FLAGS}"
env "$@" timeout 900 python bench.py $F > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "$tag FAILED"; tail -5 gpurun_out/ab_$tag.err; }
python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$tag.json").read().strip().splitlines()[-1])
s=d.get("single_scan",{})
print("$tag: value %.3e  sweep %.4f ms / %d pts  single: %.1f us/scan sweep, %.3f ms/match, iters %.2f, err %.4f" % (d["value"], d["roofline"]["avg_kernel_ms"], d["roofline"]["points_per_launch"], s.get("sweep_us_per_full_scan",0), s.get("ms_per_scanmatch",0), d["config"]["gn_iters_per_scan"], d["config"]["pose_err_vs_ground_truth_m"]["max"]))
PY
