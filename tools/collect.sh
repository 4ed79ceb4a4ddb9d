#!/bin/bash
# One GPU call that gathers the benchmark, profile and counter files kept under profiles/ for the current build:
#   tools/collect.sh <tag> [bench|pmc]   (run on the GPU box from the repo root; results under gpurun_out/<tag>/;
#                                     "bench": only the bench.py lines, e.g. after profiles/hbm_traffic.json changed)
# Afterwards, here: tools/collect_summarise.sh <tag> turns them into profiles/<tag>_* and profiles/hbm_traffic.json.
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="timeout -k 10 300 python bench.py"
if [ "${2:-}" != "pmc" ]; then   # ("pmc": only the counter / trace / CLI half, after a "bench" call)
# the headline line (with the CPU baseline leg), and the other configurations as side measurements
$B > $OUT/bench_cornell_mode1.json 2> $OUT/bench.err || exit 1
$B --mode 2 --steps 4 --warmup 1 --no-cpu-baseline --no-strong-leg > $OUT/bench_cornell_mode2.json 2>> $OUT/bench.err || exit 1
# (steps of many samples: the single kernel needs them to amortise the end of a launch -- 16 spp of 1024^2 are 4 work
#  units per resident wave -- and the streamed mode to fill its batches of up to 2^28 paths)
for m in 1 2; do
  SB=256; SH=64
  $B --no-strong-leg --scene bunny --mode $m --steps 4 --warmup 1 --spp-per-step $SB > $OUT/bench_bunny_mode$m.json 2>> $OUT/bench.err || exit 1
  $B --no-strong-leg --scene heightfield:708 --width 2048 --height 2048 --mode $m --steps 3 --warmup 1 --spp-per-step $SH > $OUT/bench_heightfield708_mode$m.json 2>> $OUT/bench.err || exit 1
  $B --no-strong-leg --no-cpu-baseline --traversal culled --scene bunny --mode $m --steps 3 --warmup 1 --spp-per-step $SB > $OUT/bench_bunny_culled_mode$m.json 2>> $OUT/bench.err || exit 1
  $B --no-strong-leg --no-cpu-baseline --traversal culled --scene heightfield:708 --width 2048 --height 2048 --mode $m --steps 2 --warmup 1 --spp-per-step $SH > $OUT/bench_heightfield708_culled_mode$m.json 2>> $OUT/bench.err || exit 1
  $B --no-strong-leg --no-cpu-baseline --bvh-build host --scene bunny --mode $m --steps 3 --warmup 1 --spp-per-step $SB > $OUT/bench_bunny_reftree_mode$m.json 2>> $OUT/bench.err || exit 1
  $B --no-strong-leg --no-cpu-baseline --bvh-build host --scene heightfield:708 --width 2048 --height 2048 --mode $m --steps 2 --warmup 1 --spp-per-step $SH > $OUT/bench_heightfield708_reftree_mode$m.json 2>> $OUT/bench.err || exit 1
done
$B --no-strong-leg --scene glass --steps 4 --warmup 1 > $OUT/bench_glass_mode1.json 2>> $OUT/bench.err || exit 1
HPT_BENCH_FORCE_DIST=1 $B --no-cpu-baseline --steps 4 --warmup 1 > $OUT/bench_cornell_mode1_rccl_1rank.json 2>> $OUT/bench.err || exit 1
[ "${2:-}" = "bench" ] && { echo collected; exit 0; }
fi
# counters and kernel traces: the headline kernel, and the streamed kernel on the two large scenes
bash tools/pmc_scene.sh ${TAG}_cornell_m1 || exit 1
bash tools/pmc_scene.sh ${TAG}_bunny_m2 --scene bunny --mode 2 --spp-per-step 256 || exit 1
bash tools/pmc_scene.sh ${TAG}_hf_m2 --scene heightfield:708 --mode 2 --spp-per-step 64 --width 2048 --height 2048 || exit 1
cd $GRAFT_REPO_ROOT
if [ -f cudapathtracer_amd/libhpt_count.so ]; then
  HPT_LIBHPT=$PWD/cudapathtracer_amd/libhpt_count.so timeout -k 10 300 python tools/count_tests.py > $OUT/executed_tests.json 2>> $OUT/bench.err || exit 1
fi
timeout -k 10 100 python tools/trace_bench.py > $OUT/intersect_microbench.txt 2>&1 || exit 1
[ -x tools/micro/gather64 ] && timeout -k 10 100 tools/micro/gather64 > $OUT/gather64.txt 2>&1
# the full configurations through the C++ CLI
mkdir -p $OUT/cli/out/gpu && cd $OUT/cli
export HPT_MODELS_DIR=$GRAFT_REPO_ROOT/models
( HPT_WIDTH=1024 HPT_HEIGHT=1024 timeout -k 10 300 $GRAFT_REPO_ROOT/build/gpu_path_tracer 4096 20 256 1 ) > $OUT/cli_config2.txt 2>&1 || exit 1
( HPT_SCENE=bunny HPT_WIDTH=1024 HPT_HEIGHT=1024 timeout -k 10 300 $GRAFT_REPO_ROOT/build/gpu_path_tracer 1024 20 256 2 ) > $OUT/cli_config3.txt 2>&1 || exit 1
( HPT_SCENE=heightfield:708 HPT_WIDTH=2048 HPT_HEIGHT=2048 timeout -k 10 600 $GRAFT_REPO_ROOT/build/gpu_path_tracer 128 20 256 2 ) > $OUT/cli_config5_128spp.txt 2>&1 || exit 1
( HPT_SCENE=heightfield:708 HPT_WIDTH=2048 HPT_HEIGHT=2048 timeout -k 10 600 $GRAFT_REPO_ROOT/build/gpu_path_tracer 1024 20 256 2 ) > $OUT/cli_config5_1024spp.txt 2>&1 || exit 1
( HPT_BVH=reference HPT_SCENE=bunny HPT_WIDTH=1024 HPT_HEIGHT=1024 timeout -k 10 300 $GRAFT_REPO_ROOT/build/gpu_path_tracer 1024 20 256 2 ) > $OUT/cli_config3_reftree.txt 2>&1 || exit 1
( HPT_GPUS=1 HPT_DEVICES=0,0 HPT_WIDTH=1024 HPT_HEIGHT=1024 timeout -k 10 300 $GRAFT_REPO_ROOT/build/gpu_path_tracer 4096 20 256 1 ) > $OUT/cli_config4_two_contexts_one_gpu.txt 2>&1 || exit 1
rm -rf $OUT/cli
echo collected
