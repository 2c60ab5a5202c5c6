#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/collect_all.sh <tag>): EVERYTHING profiles/ keeps for a round, in one call --
# the bench line and its report, kernel stats, the counter passes (collect_profiles.sh, which fails loudly when the kept kernel
# statistics do not reproduce the line), the kernels the parity tests launch, the mapping frame (steps, timeline, soak, traffic),
# the mapping node in C++, call latency, single-scan search modes, the pose graph, the neighbour-change statistics.
ulimit -c 0
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mc=/tmp/lslam_${tag}_map
mkdir -p $out $root/build
cd $root
rc=0
bash tools/collect_profiles.sh $tag || rc=1
ln -sf ${mc}.rank0.npz $root/build/_mc.rank0.npz      # the frame tools' default cache path
bash tools/profile_tests.sh $tag > $out/${tag}_profile_tests.log 2>&1
python3 tools/frame_trace.py --map-cache $mc --frames 400 2>&1 | grep -v amdgpu.ids > $out/${tag}_frame_steps.txt
( cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace -d $out/ft -o ft --output-format csv -- python3 $root/tools/frame_trace.py --map-cache $mc --frames 4 --mark > /dev/null 2>&1 )
python3 tools/frame_trace.py --timeline $(find $out/ft -name '*kernel_trace.csv' | head -1) > $out/${tag}_frame_timeline.txt 2>&1; rm -rf $out/ft
python3 tools/frame_trace.py --map-cache $mc --frames 10000 2>&1 | grep -v amdgpu.ids > $out/${tag}_frame_soak.txt
bash tools/profile_frame.sh $tag > /dev/null 2>&1
python3 tools/mapping_node_bench.py --map-cache $mc --frames 400 2>&1 | grep -v amdgpu.ids > $out/${tag}_mapping_node.txt
python3 tools/call_latency.py --map-cache $mc 2>&1 | grep -v amdgpu.ids > $out/${tag}_call_latency.txt
python3 tools/single_scan_modes.py --map-cache $mc 2>&1 | grep -v amdgpu.ids > $out/${tag}_single_scan_modes.txt
PG_PMC=1 bash tools/profile_posegraph.sh $tag > $out/${tag}_profile_pg.log 2>&1
python3 tools/nb_change_stats.py 8 10000 --map-cache $mc 2>&1 | grep -v amdgpu.ids > $out/${tag}_nb_change_stats.txt
# the odometry node (round 6): its kernels' durations over a drive of 14 sweeps, the per-query profile of its correspondence
# search, the per-sweep chain with the node's loop times
for r in 16 64; do
  ( cd /tmp && export TMPDIR=/tmp; SWEEPS=14 rocprofv3 --kernel-trace -d $out/odom_prof$r -o odom -- python3 $root/tools/bench_pipeline.py $r > /dev/null 2>&1 )
  echo "== $r rings: kernels of tools/bench_pipeline.py (14 sweeps; rocprofv3 --kernel-trace)" >> $out/${tag}_odom_kernels.txt
  python3 tools/prof_kernels.py $out/odom_prof$r/odom_results.db odom fx_ ms_ solve_kernel >> $out/${tag}_odom_kernels.txt 2>&1
  rm -rf $out/odom_prof$r
  python3 tools/odom_search_stats.py $r 2>&1 | grep -v amdgpu.ids >> $out/${tag}_odom_search_stats.txt
  DETAIL=1 SWEEPS=14 python3 tools/bench_pipeline.py $r 2>&1 | grep -v amdgpu.ids >> $out/${tag}_odom_chain.txt
done
build/ubench_valu > $out/${tag}_ubench_valu.json 2>/dev/null
ls -la $out | grep ${tag}_ | awk '{print $5, $9}'
exit $rc
