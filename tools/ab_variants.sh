#!/bin/bash
ulimit -c 0  # a GPU memory fault must not leave a core dump that fills the box's disk
# On the GPU box: the headline command through several builds of the library (tools/build_variant.sh), interleaved, N rounds:
#   tools/ab_variants.sh <rounds> <steps> name1 name2 ...     ("base" = the in-tree library)
rounds=$1; steps=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    lib=$root/build/exp/$v.so; [ "$v" = base ] && lib=$root/the-cooper-mapper_amd/liblslam_hip.so
    LSLAM_ALLOW_EXPERIMENT_BUILD=1 LSLAM_LIB=$lib timeout 600 python bench.py --headline-only --steps $steps --warmup 1 --map-cache /tmp/ab_map > gpurun_out/abv_$v.json 2> gpurun_out/abv_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/abv_$v.err; continue; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/abv_$v.json").read().strip().splitlines()[-1])
print("round $r %-14s value %.4e  ms/step %.3f  sweep %.4f ms  launches %d iters %.2f unproven %.4f  pose_err %.4f conv %d poses %s" % ("$v", d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["launches_timed"], d["config"]["gn_iters_per_scan"], d["grid_sweep"]["share_left_to_the_tree_search"], d["config"]["pose_err_vs_ground_truth_m"], d["config"]["converged_scans"], d["config"].get("poses_crc32")))
PY
  done
done
