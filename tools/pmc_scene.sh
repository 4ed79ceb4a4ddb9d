#!/bin/bash
# PMC passes (each in its own rocprofv3 process) + kernel trace over one bench configuration:
#   tools/pmc_scene.sh <tag> <bench args...>     (run on the GPU box from the repo root)
# Results under gpurun_out/pmc_<tag>/; tools/pmc_to_json.py turns them into a profiles/hbm_traffic.json entry.
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT   # (results of an earlier collection would be averaged in)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="$*"
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg $BENCH_ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
pass p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM
pass p3 FETCH_SIZE
pass p4 WRITE_SIZE
pass p5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass p6 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum
pass p7 TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strong-leg $BENCH_ARGS > $OUT/trace.log 2>&1 || { echo "trace failed"; exit 1; }
echo ok $TAG
