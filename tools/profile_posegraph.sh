#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of the pose-graph LM run (tools/bench_posegraph.py) -> gpurun_out/<tag>_pg_kernel_stats.csv
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ITERS=1000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_pgstats -o s -- python3 $root/tools/bench_posegraph.py > $out/${tag}_pgstats.log 2>&1
cp $out/${tag}_pgstats/s_kernel_stats.csv $out/${tag}_pg_kernel_stats.csv
rm -rf $out/${tag}_pgstats
grep "LM iters" $out/${tag}_pgstats.log
head -14 $out/${tag}_pg_kernel_stats.csv | cut -c1-160
# counter passes of the same run (one set per pass, no tracing domains): memory traffic and issue / wait counters of the
# persistent PCG kernel, summed over a launch (= one damped solve), mean over the launches
if [ -n "$PG_PMC" ]; then
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    ITERS=1000 timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_pgpmc/p$i -o p -- python3 $root/tools/bench_posegraph.py > $out/${tag}_pgpmc_p$i.log 2>&1
  done
  python3 $root/tools/summarize_pmc.py pg_pcg_persistent_kernel $out/${tag}_pg_pmc.csv $out/${tag}_pgpmc/p1 $out/${tag}_pgpmc/p2 $out/${tag}_pgpmc/p3 $out/${tag}_pgpmc/p4 | sed 's/bench.py --headline-only --steps 6 --warmup 1/tools\/bench_posegraph.py/' > /dev/null
  sed -i 's/python3 bench.py --headline-only --steps 6 --warmup 1/python3 tools\/bench_posegraph.py (ITERS=1000)/' $out/${tag}_pg_pmc.csv
  rm -rf $out/${tag}_pgpmc
  cat $out/${tag}_pg_pmc.csv
fi
