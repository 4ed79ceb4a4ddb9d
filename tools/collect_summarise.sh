#!/bin/bash
# gpurun_out/<tag>/ (tools/collect.sh) -> profiles/<tag>_* and the entries of profiles/hbm_traffic.json
set -e
TAG=$1
cd "$(dirname "$0")/.."
G=gpurun_out/$TAG
P=profiles
# nothing is overwritten unless the collection is there
for d in $G gpurun_out/pmc_${TAG}_cornell_m1 gpurun_out/pmc_${TAG}_bunny_m2 gpurun_out/pmc_${TAG}_hf_m2; do [ -d $d ] || { echo "missing $d: run tools/collect.sh $TAG on the GPU box first"; exit 1; }; done
[ -f $G/bench_cornell_mode1.json ] || { echo "missing $G/bench_cornell_mode1.json"; exit 1; }
for f in $G/bench_*.json $G/intersect_microbench.txt $G/gather64.txt $G/cli_config*.txt; do [ -f $f ] && cp $f $P/${TAG}_$(basename $f); done
[ -f $G/executed_tests.json ] && cp $G/executed_tests.json $P/executed_tests.json
python tools/pmc_to_json.py gpurun_out/pmc_${TAG}_cornell_m1 k_render_mega cornell_mode1_1024x1024_d20_spp1024 $P/${TAG}_cornell_mode1_pmc_summary.txt > /dev/null
python tools/pmc_to_json.py gpurun_out/pmc_${TAG}_bunny_m2 k_wf_bounces bunny_mode2_1024x1024_d20_spp256 $P/${TAG}_bunny_mode2_pmc_summary.txt > /dev/null
python tools/pmc_to_json.py gpurun_out/pmc_${TAG}_hf_m2 k_wf_bounces heightfield:708_mode2_2048x2048_d20_spp64 $P/${TAG}_heightfield708_mode2_pmc_summary.txt > /dev/null
for s in cornell_m1 bunny_m2 hf_m2; do
  f=$(ls -t gpurun_out/pmc_${TAG}_$s/trace/*/*kernel_stats.csv | head -1)   # the newest (the directory may hold earlier collections)
  cp $f $P/${TAG}_${s}_kernel_stats.csv
done
ls $P | grep "^$TAG" 
