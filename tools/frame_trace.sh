set -e
python tools/frame_trace.py --map-cache build/_mc --frames 30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/ft -o ft --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame_trace.py --map-cache $GRAFT_REPO_ROOT/build/_mc --frames 4 --mark > /dev/null 2>&1 || true
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/ft -name '*kernel_trace.csv' | head -1)
python tools/frame_trace.py --timeline $f
