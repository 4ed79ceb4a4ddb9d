#!/bin/bash
# Rendering speed on the host-built (reference) tree against the device-built one, both large scenes, both modes
# (run on the GPU box from the repo root):   tools/ab_bvh_build.sh > gpurun_out/ab_bvh.txt
P='import json,sys; d=json.load(sys.stdin); print(sys.argv[1], d["config"]["bvh_build"], "mode", sys.argv[2], "%.0f Mrays/s" % d["value"], "device build ms:", d["config"]["gpu_bvh_build_ms"])'
for b in host gpu; do
  for m in 2 1; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --scene bunny --mode $m --spp-per-step 16 --steps 3 --warmup 1 --bvh-build $b | python -c "$P" bunny $m || exit 1
    timeout -k 10 200 python bench.py --no-cpu-baseline --scene heightfield:708 --width 2048 --height 2048 --mode $m --spp-per-step 4 --steps 2 --warmup 1 --bvh-build $b | python -c "$P" heightfield:708 $m || exit 1
  done
done
