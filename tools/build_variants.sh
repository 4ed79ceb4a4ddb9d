#!/bin/bash
# Build differently configured kernel libraries for A/B runs on the GPU box:
#   tools/build_variants.sh tag1:"-DX=1 -DY=2" tag2:"..."      -> build/tune/libhpt_<tag>.so
# Run one with HPT_LIBHPT=$PWD/build/tune/libhpt_<tag>.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/tune
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Wno-unused-function"
pids=()
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -shared -o build/tune/libhpt_$tag.so cudapathtracer_amd/csrc/hpt_core.hip cudapathtracer_amd/csrc/hpt_bvh_build.hip &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls -la build/tune
