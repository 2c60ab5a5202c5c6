#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/profile_tests.sh <tag>): rocprofv3 kernel stats of the GPU parity tests that hold the
# bench's kernel instantiations against the oracle (tests/test_gpu_stack_shapes.py) -> gpurun_out/<tag>_tests_kernel_stats.csv.
# The summary shows which sweep kernels the tests really launched: sweep_kernel<256,true,false,12,...> (the one bench.py times),
# its certificate pass and sweep_queue_kernel among them.
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_tests_stats -o s -- python3 -m pytest $root/tests/test_gpu_stack_shapes.py -m gpu -q -p no:cacheprovider > $out/${tag}_tests_stats.log 2>&1
cp $out/${tag}_tests_stats/s_kernel_stats.csv $out/${tag}_tests_kernel_stats.csv
rm -rf $out/${tag}_tests_stats
grep -a "passed\|failed" $out/${tag}_tests_stats.log | tail -1
grep -a "sweep_\|cert_plan" $out/${tag}_tests_kernel_stats.csv | cut -c1-200
