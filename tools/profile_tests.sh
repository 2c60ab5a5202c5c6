#!/bin/bash
# Run on the GPU box: which kernels the parity tests really launch -- rocprofv3 --kernel-trace --stats around the grid / stack-shape /
# map tests -> gpurun_out/<tag>_tests_kernel_stats.csv (the instantiations the bench times must be in it)
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_tstats -o s -- python3 -m pytest $root/tests/test_gpu_grid.py $root/tests/test_gpu_stack_shapes.py $root/tests/test_gpu_fmap.py -q -m gpu -p no:cacheprovider > $out/${tag}_tstats.log 2>&1
cp $(find $out/${tag}_tstats -name '*kernel_stats.csv' | head -1) $out/${tag}_tests_kernel_stats.csv
rm -rf $out/${tag}_tstats
tail -3 $out/${tag}_tstats.log
grep -c . $out/${tag}_tests_kernel_stats.csv
