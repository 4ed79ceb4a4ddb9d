#!/bin/bash
# One GPU call that gathers the benchmark, profile and counter files kept under profiles/ for the current build:
#   tools/collect.sh <tag>     (run on the GPU box from the repo root; results under gpurun_out/<tag>/)
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mode1 -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/prof_mode1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mode2 -- python3 /root/repo/bench.py --no-cpu-baseline --mode 2 --steps 4 --warmup 1 --spp-per-step 16 > $OUT/prof_mode2.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh ${TAG}_pmc > /dev/null 2>&1 || exit 1
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --mode 2 --steps 3 --warmup 1 --spp-per-step 16 > $OUT/bench_mode2.json 2>> $OUT/bench.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --scene bunny --steps 3 --warmup 1 --spp-per-step 16 > $OUT/bench_bunny.json 2>> $OUT/bench.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --scene bunny --mode 2 --steps 3 --warmup 1 --spp-per-step 16 > $OUT/bench_bunny_mode2.json 2>> $OUT/bench.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --scene heightfield:708 --steps 2 --warmup 1 --spp-per-step 4 > $OUT/bench_heightfield.json 2>> $OUT/bench.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --scene heightfield:708 --mode 2 --steps 2 --warmup 1 --spp-per-step 4 > $OUT/bench_heightfield_mode2.json 2>> $OUT/bench.err || exit 1
for m in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --traversal culled --scene bunny --mode $m --steps 3 --warmup 1 --spp-per-step 16 > $OUT/bench_bunny_culled_mode$m.json 2>> $OUT/bench.err || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --traversal culled --scene heightfield:708 --mode $m --steps 2 --warmup 1 --spp-per-step 4 > $OUT/bench_heightfield_culled_mode$m.json 2>> $OUT/bench.err || exit 1
done
bash tools/pmc_mem.sh ${TAG}_bunny --scene bunny --spp-per-step 16 > /dev/null 2>&1 || exit 1
bash tools/pmc_mem.sh ${TAG}_hf --scene heightfield:708 --spp-per-step 4 > /dev/null 2>&1 || exit 1
timeout -k 10 100 python tools/cull_probe.py > $OUT/cull_probe.txt 2>&1 || exit 1
timeout -k 10 100 python tools/trace_bench.py > $OUT/trace_bench.txt 2>&1 || exit 1
if [ -f build/libhpt_prof.so ]; then
  HPT_LIBHPT=$PWD/build/libhpt_prof.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp-per-step 64 2>&1 | grep PROF > $OUT/section_profile.txt
fi
echo collected
