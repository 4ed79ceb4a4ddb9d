#!/bin/bash
# A/B of compiler-flag builds made by tools/build_variants.sh (tags base, ilp, relaxed, nounroll, maxocc) on the three bench scenes
L="build/tune/libhpt_base.so build/tune/libhpt_ilp.so build/tune/libhpt_relaxed.so build/tune/libhpt_nounroll.so build/tune/libhpt_maxocc.so"
echo "== cornell m1"; tools/ab.sh "--steps 4 --warmup 1" $L || exit 1
echo "== bunny m2"; tools/ab.sh "--scene bunny --mode 2 --steps 3 --warmup 1" $L || exit 1
echo "== hf m2"; tools/ab.sh "--scene heightfield:708 --width 2048 --height 2048 --mode 2 --spp-per-step 64 --steps 2 --warmup 1" $L || exit 1
