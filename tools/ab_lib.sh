#!/bin/bash
# A/B of differently built kernel libraries on the large scenes, streamed mode (run on the GPU box from the repo root):
#   tools/ab_lib.sh <tag> <lib.so | default> [more libs...]
set -u
TAG=$1; shift
mkdir -p gpurun_out
for LIB in "$@"; do
  name=$(basename $LIB .so)
  if [ "$LIB" != default ]; then export HPT_LIBHPT=$PWD/$LIB; else unset HPT_LIBHPT; fi
  for sc in "bunny --width 1024 --height 1024 --spp-per-step 256" "heightfield:708 --width 2048 --height 2048 --spp-per-step 64"; do
    s=${sc%% *}
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-strong-leg --warmup 1 --steps 3 --mode ${MODE:-2} --scene $sc > gpurun_out/${TAG}_${name}_${s%%:*}.json 2> gpurun_out/${TAG}_${name}_${s%%:*}.err || { echo "$name $s failed"; tail -3 gpurun_out/${TAG}_${name}_${s%%:*}.err; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('gpurun_out/${TAG}_${name}_${s%%:*}.json'))
print('%-16s %-16s %8.0f Mrays/s %8.1f ms/launch' % ('$name', '$s', d['value'], d['roofline']['kernel_ms']))"
  done
done
