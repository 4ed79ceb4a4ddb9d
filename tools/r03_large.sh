#!/bin/bash
# Side measurements of the large-scene configurations (run on the GPU box from the repo root):
#   tools/r03_large.sh <tag> [extra bench args...]
# bunny 1024x1024 and the 1.0 M-triangle height field 2048x2048, both modes, default tree; results under gpurun_out/<tag>_*.json
set -u
TAG=$1; shift
mkdir -p gpurun_out
run() {
  name=$1; shift
  timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-strong-leg --warmup 1 "$@" > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err || { echo "$name failed"; tail -3 gpurun_out/${TAG}_$name.err; return 1; }
  python3 - gpurun_out/${TAG}_$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("%-28s %8.0f Mrays/s  %7.1f ms/launch  %s  bvh %s" % (sys.argv[1].split("/")[-1], d["value"], r["kernel_ms"], r["kernel"], d["config"]["bvh_build"]))
PY
}
run bunny_m2 --scene bunny --mode 2 --spp-per-step 256 --steps 3 "$@" &&
run bunny_m1 --scene bunny --mode 1 --spp-per-step 256 --steps 3 "$@" &&
run hf_m2 --scene heightfield:708 --width 2048 --height 2048 --mode 2 --spp-per-step 64 --steps 3 "$@" &&
run hf_m1 --scene heightfield:708 --width 2048 --height 2048 --mode 1 --spp-per-step 64 --steps 3 "$@"
