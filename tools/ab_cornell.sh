#!/bin/bash
# Bench every library under build/tune/ on the Cornell box, both modes (run on the GPU box from the repo root):
#   tools/ab_cornell.sh [extra bench args] > gpurun_out/abc.txt
for so in build/tune/libhpt_*.so; do
  tag=$(basename $so .so); tag=${tag#libhpt_}
  a=$(HPT_LIBHPT=$PWD/$so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" | python -c "import json,sys; print('%.0f' % json.load(sys.stdin)['value'])") || exit 1
  b=$(HPT_LIBHPT=$PWD/$so timeout -k 10 200 python bench.py --no-cpu-baseline --mode 2 --steps 4 --warmup 1 "$@" | python -c "import json,sys; print('%.0f' % json.load(sys.stdin)['value'])") || exit 1
  echo "$tag cornell mode1 $a mode2 $b"
done
