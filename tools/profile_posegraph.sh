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
