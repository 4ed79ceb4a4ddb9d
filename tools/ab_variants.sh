#!/bin/bash
# Bench every library under build/tune/ on the large scenes (run on the GPU box from the repo root):
#   tools/ab_variants.sh [mode] [extra bench args] > gpurun_out/ab.txt
MODE=${1:-2}; shift
for so in build/tune/libhpt_*.so; do
  tag=$(basename $so .so); tag=${tag#libhpt_}
  b=$(HPT_LIBHPT=$PWD/$so timeout -k 10 200 python bench.py --no-cpu-baseline --scene bunny --mode $MODE --spp-per-step 16 --steps 3 --warmup 1 "$@" | python -c "import json,sys; print('%.0f' % json.load(sys.stdin)['value'])") || exit 1
  h=$(HPT_LIBHPT=$PWD/$so timeout -k 10 200 python bench.py --no-cpu-baseline --scene heightfield:708 --width 2048 --height 2048 --mode $MODE --spp-per-step 4 --steps 2 --warmup 1 "$@" | python -c "import json,sys; print('%.0f' % json.load(sys.stdin)['value'])") || exit 1
  echo "$tag mode$MODE bunny $b heightfield $h"
done
