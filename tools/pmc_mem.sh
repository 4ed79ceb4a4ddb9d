#!/bin/bash
# Memory-side PMC passes for one bench configuration:  tools/pmc_mem.sh <tag> <bench args...>
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
BENCH_ARGS="$*"
pass m1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass m2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum
pass m3 TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum
pass m4 FETCH_SIZE
pass m5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
echo ok
