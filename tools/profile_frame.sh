#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/profile_frame.sh <tag>): HBM-side traffic of the kernels of a mapping frame, per kernel
# name and launch -- the cell-grid build (grid_*), the map rebuild (fm_*), the frame's sweeps -- from two rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; one counter per pass, kernel trace only) around tools/frame_trace.py.  -> gpurun_out/<tag>_frame_pmc_by_kernel.csv
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_fpmc/p$i -o p -- python3 $root/tools/frame_trace.py --map-cache $root/build/_mc --frames 10 --mark > $out/${tag}_fpmc_p$i.log 2>&1
done
python3 $root/tools/summarize_pmc_by_name.py $out/${tag}_frame_pmc_by_kernel.csv "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> --output-format csv -- python3 tools/frame_trace.py --frames 10 (one pass per counter): the kernels of a mapping frame, per kernel name; KiB per launch as reported (HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, MI355X_MICROARCH.md)" grid_,fm_,sweep_grid,sweep_wide,sweep_queue,fx_ring,fx_compact $out/${tag}_fpmc/p1 $out/${tag}_fpmc/p2 > /dev/null
rm -rf $out/${tag}_fpmc
cat $out/${tag}_frame_pmc_by_kernel.csv
