#!/bin/bash
# A/B of the host tree builders on the large scenes (run on the GPU box from the repo root):
#   tools/ab_sah.sh > gpurun_out/ab_sah.txt
for scene in bunny heightfield:708; do
  for m in 2 1; do
    for rep in 1 2; do
      a=$(timeout -k 10 300 python bench.py --no-cpu-baseline --scene $scene --mode $m --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; print('%.0f' % json.load(sys.stdin)['value'])") || exit 1
      b=$(timeout -k 10 300 python bench.py --no-cpu-baseline --bvh-build sah --scene $scene --mode $m --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; print('%.0f' % json.load(sys.stdin)['value'])") || exit 1
      echo "$scene mode $m: median split $a  SAH $b Mrays/s"
    done
  done
done
