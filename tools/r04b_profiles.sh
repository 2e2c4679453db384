#!/bin/bash
# Round-4 evidence run after the remaining-work priority rule (pagk_prio.h): rocprofv3 profiles (kernel trace + separate --pmc passes)
# of the bench.py headline and of the pipelined 4-wave kernel on configs[1] / [2]; then the full bench.py line.
# Summaries: python tools/save_profile.py r04b_bench ; python tools/save_profile_variant.py r04b_a_cfg1 ; ... r04b_a_cfg2
set -o pipefail
mkdir -p gpurun_out/r04bp
for spec in "r04b_a_cfg1 1 1000 0" "r04b_a_cfg2 2 2000 0"; do
  set -- $spec
  timeout -k 10 240 bash tools/profile_variant.sh $1 $2 $3 $4 > gpurun_out/r04bp/$1.log 2>&1 || echo "profile $1 failed" >> gpurun_out/r04bp/failed.txt
  echo "profiled $1"
done
timeout -k 10 300 bash tools/profile.sh r04b_bench > gpurun_out/r04bp/profile_bench.log 2>&1 || echo "profile bench failed" >> gpurun_out/r04bp/failed.txt
echo "profiled bench"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_r04b_bench.json 2> gpurun_out/r04bp/bench.err || echo "bench failed" >> gpurun_out/r04bp/failed.txt
tail -c 300 gpurun_out/r04bp/bench.err
head -c 400 gpurun_out/bench_r04b_bench.json
if [ -f gpurun_out/r04bp/failed.txt ]; then cat gpurun_out/r04bp/failed.txt; fi
