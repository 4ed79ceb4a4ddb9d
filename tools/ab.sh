#!/bin/bash
# A/B several builds of libhpt.so in one GPU call:  tools/ab.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
for lib in "$@"; do
  r=$(HPT_LIBHPT=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline $ARGS 2>&1 | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' ')
  echo "$lib: $r"
done
