#!/bin/bash
run() { r=$(env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --spp-per-step 1024 --steps 2 --warmup 1 2>&1 | grep -o '"value": [0-9.]*') || exit 1; echo "$* : $r"; }
run HPT_SPP_PER_LAUNCH=256 HPT_SPP_PER_UNIT=16
run HPT_SPP_PER_LAUNCH=1024 HPT_SPP_PER_UNIT=16
run HPT_SPP_PER_LAUNCH=256 HPT_SPP_PER_UNIT=8
run HPT_SPP_PER_LAUNCH=256 HPT_SPP_PER_UNIT=32
run HPT_SPP_PER_LAUNCH=1024 HPT_SPP_PER_UNIT=32
run HPT_SPP_PER_LAUNCH=1024 HPT_SPP_PER_UNIT=64
run HPT_SPP_PER_LAUNCH=256 HPT_SPP_PER_UNIT=16
