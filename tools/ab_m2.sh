#!/bin/bash
# A/B of kernel-library builds on the two large scenes, streamed mode:  tools/ab_m2.sh lib1.so lib2.so ...
echo "== bunny m2"; tools/ab.sh "--scene bunny --mode 2 --steps 2 --warmup 1" "$@" || exit 1
echo "== hf m2"; tools/ab.sh "--scene heightfield:708 --width 2048 --height 2048 --mode 2 --spp-per-step 64 --steps 2 --warmup 1" "$@" || exit 1
