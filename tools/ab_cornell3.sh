#!/bin/bash
# Cornell 1024x1024 depth 20 in both modes with one library (run on the GPU box from the repo root):
#   tools/ab_cornell3.sh <tag> [lib.so]
set -u
TAG=$1; LIB=${2:-default}
[ "$LIB" != default ] && export HPT_LIBHPT=$PWD/$LIB
for m in 1 2; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-strong-leg --warmup 1 --steps 3 --mode $m > gpurun_out/${TAG}_cornell_m$m.json 2> gpurun_out/${TAG}_cornell_m$m.err || { echo "cornell mode $m failed"; tail -3 gpurun_out/${TAG}_cornell_m$m.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/${TAG}_cornell_m$m.json'))
print('%-10s cornell mode $m %8.0f Mrays/s %8.2f ms/launch' % ('$TAG', d['value'], d['roofline']['kernel_ms']))"
done
