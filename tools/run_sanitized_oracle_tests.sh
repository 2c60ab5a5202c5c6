#!/bin/bash
# The C oracles under AddressSanitizer + UndefinedBehaviorSanitizer, scripted: builds oracle/_asan/*.so and
# runs the CPU test suite's oracle tests against them (LSLAM_ORACLE_SANITIZE=1 makes the loaders take the sanitized builds;
# the sanitizer runtimes must be in the process before python loads them, hence LD_PRELOAD).  No GPU, no HIP library involved:
# the sanitizers see the restatement's own loops -- kd-tree build / search, fits, the Gauss-Newton loop, map maintenance,
# feature extraction, the pose-graph Cholesky.
#   tools/run_sanitized_oracle_tests.sh            # exits non-zero on the first report
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C "$root/oracle" asan > /dev/null
asan=$(gcc -print-file-name=libasan.so)
ubsan=$(gcc -print-file-name=libubsan.so)
cd "$root"
LD_PRELOAD="$asan:$ubsan" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  LSLAM_ORACLE_SANITIZE=1 python3 -m pytest -q -x -s -m "not gpu" -p no:cacheprovider \
  tests/test_oracle_kdtree.py tests/test_oracle_math.py tests/test_oracle_fmap.py tests/test_oracle_features.py \
  tests/test_oracle_stereo.py tests/test_oracle_posegraph_c.py tests/test_threshold_parity.py tests/test_certificate_property.py "$@"
