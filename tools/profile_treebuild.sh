#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/profile_treebuild.sh <tag>): kernel stats and PMC passes of the kd-tree build of the
# bench's surround (tools/bench_treebuild.py; needs build/surround_cache.npz).  Results: gpurun_out/<tag>_tree_*.
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export REPS=12
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_tree_stats -o s -- python3 $root/tools/bench_treebuild.py > $out/${tag}_tree_stats.log 2>&1
cp $out/${tag}_tree_stats/s_kernel_stats.csv $out/${tag}_tree_kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_tree_pmc/p$i -o p -- python3 $root/tools/bench_treebuild.py > $out/${tag}_tree_pmc_p$i.log 2>&1
done
python3 $root/tools/summarize_pmc_kernels.py $out/${tag}_tree_pmc.csv $REPS lv_,kd_ $out/${tag}_tree_pmc/p1 $out/${tag}_tree_pmc/p2 $out/${tag}_tree_pmc/p3 > /dev/null
rm -rf $out/${tag}_tree_pmc $out/${tag}_tree_stats
tail -2 $out/${tag}_tree_stats.log
head -40 $out/${tag}_tree_pmc.csv
