#!/bin/bash
# PMC passes of the sweep kernel with ONE scan in flight (bench.py --batch 1): run on the GPU box.
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py --batch 1 --steps 10 --warmup 2 --no-cpu-baseline --no-pose-graph --no-single --no-mapping-frame --no-joint-stereo"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_pmc1/p$i -o p -- python3 $B > $out/${tag}_pmc1_p$i.log 2>&1
done
python3 $root/tools/summarize_pmc.py sweep_kernel $out/${tag}_pmc_sweep_batch1.csv $out/${tag}_pmc1/p1 $out/${tag}_pmc1/p2 $out/${tag}_pmc1/p3 $out/${tag}_pmc1/p4 $out/${tag}_pmc1/p5 > /dev/null
sed -i 's/--no-cpu-baseline --no-pose-graph --no-single --no-mapping-frame   (one pass/--batch 1 --no-cpu-baseline --no-pose-graph --no-single --no-mapping-frame --no-joint-stereo   (one pass/' $out/${tag}_pmc_sweep_batch1.csv
rm -rf $out/${tag}_pmc1
