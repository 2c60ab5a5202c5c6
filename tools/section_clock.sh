#!/bin/bash
ulimit -c 0
# On the GPU box: where a wavefront of the grid sweep's pass 1 spends its life.  Builds the library with s_memtime readings at
# the section ends of sweep_grid_kernel (-DLSLAM_EXP_SECTION_CLOCK: same results, nine scalar accumulators per wavefront, one
# workgroup in sixteen adds them to a table), runs the headline command through it and through the in-tree build, interleaved,
# and prints the shares.  tools/section_clock.sh [rounds] [steps]
rounds=${1:-2}; steps=${2:-6}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
tools/build_variant.sh secclock "-DLSLAM_EXP_SECTION_CLOCK" || exit 1
python bench.py --headline-only --steps 1 --warmup 1 --map-cache /tmp/ab_map > /dev/null 2>&1
tools/ab_variants.sh $rounds $steps base secclock | tee gpurun_out/section_clock.txt
python - <<PY | tee -a gpurun_out/section_clock.txt
import json
d = json.loads(open("gpurun_out/abv_secclock.json").read().strip().splitlines()[-1])
b = json.loads(open("gpurun_out/abv_base.json").read().strip().splitlines()[-1])
sc = d["section_clock"]
print("sweep_grid_kernel, %d wavefronts reporting, %.0f s_memtime ticks per wavefront; sweep %.4f ms with the clocks, %.4f without" %
      (sc["wavefronts"], sc["ticks_per_wavefront"], d["roofline"]["avg_kernel_ms"], b["roofline"]["avg_kernel_ms"]))
print("candidate loop: %.1f candidates per point (lanes without a point included), %.1f rounds per wavefront: %.3f of the lane-rounds carry a candidate" %
      (sc["candidates_per_point"], sc["loop_rounds_per_wavefront"], sc["loop_lane_use"]))
for k, v in sc["share"].items():
    print("  %5.1f %%  %s" % (100 * v, k))
json.dump({"with_clocks": d, "base": b}, open("gpurun_out/section_clock.json", "w"), indent=1)
PY
