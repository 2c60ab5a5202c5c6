#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/profile_grid.sh <tag> [search] [extra bench args]): kernel-trace stats and the VALU / lane counters
# of `bench.py --headline-only --search <search>` -> gpurun_out/<tag>_<search>_{kernel_stats.csv,pmc.csv,line.json}
tag=${1:-r04}
search=${2:-grid}
shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
H="$root/bench.py --headline-only --steps 6 --warmup 1 --search $search --map-cache /tmp/lslam_${tag}_map $@"
timeout 600 python3 $H > $out/${tag}_${search}_line.json 2> $out/${tag}_${search}_line.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${search}_hs -o s -- python3 $H > $out/${tag}_${search}_hs.log 2>&1
cp $out/${tag}_${search}_hs/s_kernel_stats.csv $out/${tag}_${search}_kernel_stats.csv
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
  "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_${search}_pmc/p$i -o p -- python3 $H > $out/${tag}_${search}_pmc_p$i.log 2>&1
done
first=sweep_kernel; [ "$search" = grid ] && first=sweep_grid_kernel
python3 $root/tools/summarize_pmc.py $first,sweep_queue_kernel,cert_plan_kernel $out/${tag}_${search}_pmc.csv $out/${tag}_${search}_pmc/p1 $out/${tag}_${search}_pmc/p2 > /dev/null
for k in $first sweep_queue_kernel; do   # and each kernel on its own (mean per launch of THAT kernel)
  python3 $root/tools/summarize_pmc.py $k $out/${tag}_${search}_pmc_$k.csv $out/${tag}_${search}_pmc/p1 $out/${tag}_${search}_pmc/p2 > /dev/null
done
rm -rf $out/${tag}_${search}_pmc $out/${tag}_${search}_hs
