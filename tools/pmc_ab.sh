#!/bin/bash
# One SQ counter pass per library under build/tune/ on one bench configuration (run on the GPU box from the repo root):
#   tools/pmc_ab.sh <bench args...>   -> gpurun_out/pmc_ab/<tag>/ + a summary on stdout
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab
mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for so in $ROOT/build/tune/libhpt_*.so; do
  tag=$(basename $so .so); tag=${tag#libhpt_}
  export HPT_LIBHPT=$so
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/$tag -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $OUT/$tag.log; exit 1; }
  python3 - $OUT/$tag $tag <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        tot[k]['_n_' + r['Counter_Name']] += 1
for k, c in tot.items():
    if c.get('SQ_INSTS_VALU', 0) < 1e9: continue
    n = c['_n_SQ_INSTS_VALU']
    print(tag, k, 'launches', int(n), 'VALU %.2fG SALU %.2fG LDS %.2fG active %.3f wavecycles %.3gG' % (c['SQ_INSTS_VALU'] / n / 1e9, c['SQ_INSTS_SALU'] / n / 1e9, c['SQ_INSTS_LDS'] / n / 1e9, c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_INSTS_VALU']), c['SQ_WAVE_CYCLES'] / n / 1e9))
PY
done
