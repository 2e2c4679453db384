#!/bin/bash
# Round-3 evidence run (through gpurun): one rocprofv3 profile per tracking variant on the BASELINE config it serves
# (tools/profile_variant.sh), the bench.py profile of the headline (tools/profile.sh) and the full bench.py line.
# Summaries: python tools/save_profile_variant.py <tag> for the variant tags, python tools/save_profile.py r03_bench.
set -o pipefail
mkdir -p gpurun_out/r03p
SPECS=${PAGK_PROFILE_SPECS:-"r03_a_cfg1:1:1000:0 r03_b_cfg4:4:4000:2 r03_d_cfg3:3:20000:5 r03_f_cfg3:3:20000:7 r03_d_cfg2x:2:2000:0"}
for spec in $SPECS; do
  spec=${spec//:/ }
  set -- $spec
  timeout -k 10 240 bash tools/profile_variant.sh $1 $2 $3 $4 > gpurun_out/r03p/$1.log 2>&1 || echo "profile $1 failed" >> gpurun_out/r03p/failed.txt
  echo "profiled $1"
done
timeout -k 10 300 bash tools/profile.sh r03_bench > gpurun_out/r03p/profile_bench.log 2>&1 || echo "profile bench failed" >> gpurun_out/r03p/failed.txt
timeout -k 10 500 python3 bench.py > gpurun_out/bench_r03_bench.json 2> gpurun_out/r03p/bench.err || echo "bench failed" >> gpurun_out/r03p/failed.txt
tail -c 400 gpurun_out/r03p/bench.err
head -c 600 gpurun_out/bench_r03_bench.json
if [ -f gpurun_out/r03p/failed.txt ]; then cat gpurun_out/r03p/failed.txt; fi
