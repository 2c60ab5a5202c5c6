#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of one of the tools/ scripts -> gpurun_out/<tag>_kernel_stats.csv
#   bash tools/profile_tool.sh <tag> tools/bench_features.py [args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- python3 $root/"$@" > $out/${tag}_stats.log 2>&1
cp $out/${tag}_stats/s_kernel_stats.csv $out/${tag}_kernel_stats.csv
rm -rf $out/${tag}_stats
tail -3 $out/${tag}_stats.log
head -${TOP:-16} $out/${tag}_kernel_stats.csv | cut -c1-170
