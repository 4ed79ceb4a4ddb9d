#!/bin/bash
# PMC passes over a short bench run (each pass in its own rocprofv3 process).
#   tools/pmc.sh <tag> [bench args...]      (run on the GPU box from the repo root)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 ${PMC_SCRIPT:-/root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline} $BENCH_ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
BENCH_ARGS="$*"
pass p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM
pass p3 FETCH_SIZE
pass p4 WRITE_SIZE
pass p5 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
find $OUT -name "*counter_collection.csv" | head
