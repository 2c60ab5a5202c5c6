#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/collect_profiles.sh <tag>): bench line, rocprofv3 kernel
# stats and the PMC passes the roofline object cites.  Results land in gpurun_out/<tag>_*.
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pose-graph --no-single --no-mapping-frame --no-joint-stereo"
timeout 900 python3 $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/${tag}_stats.log 2>&1
cp $out/${tag}_stats/s_kernel_stats.csv $out/${tag}_kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_pmc/p$i -o p -- python3 $B > $out/${tag}_pmc_p$i.log 2>&1
done
python3 $root/tools/summarize_pmc.py sweep_kernel $out/${tag}_pmc_sweep_batch8.csv $out/${tag}_pmc/p1 $out/${tag}_pmc/p2 $out/${tag}_pmc/p3 $out/${tag}_pmc/p4 $out/${tag}_pmc/p5 > /dev/null
rm -rf $out/${tag}_pmc/*/*/*_agent_info.csv
du -sh $out | tail -1
