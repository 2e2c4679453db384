#!/bin/bash
# Round-4 evidence run of the final build (priority rule, ds_write flags, runtime_env): rocprofv3 profiles (kernel trace + separate --pmc passes)
# of the bench.py headline and of the pipelined 4-wave kernel on configs[1] / [2]; then the full bench.py line.
# Summaries: python tools/save_profile.py r04c_bench ; python tools/save_profile_variant.py r04c_a_cfg1 ; ... r04c_a_cfg2
set -o pipefail
mkdir -p gpurun_out/r04cp
for spec in "r04c_a_cfg1 1 1000 0" "r04c_a_cfg2 2 2000 0"; do
  set -- $spec
  timeout -k 10 240 bash tools/profile_variant.sh $1 $2 $3 $4 > gpurun_out/r04cp/$1.log 2>&1 || echo "profile $1 failed" >> gpurun_out/r04cp/failed.txt
  echo "profiled $1"
done
timeout -k 10 300 bash tools/profile.sh r04c_bench > gpurun_out/r04cp/profile_bench.log 2>&1 || echo "profile bench failed" >> gpurun_out/r04cp/failed.txt
echo "profiled bench"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_r04c_bench.json 2> gpurun_out/r04cp/bench.err || echo "bench failed" >> gpurun_out/r04cp/failed.txt
tail -c 300 gpurun_out/r04cp/bench.err
head -c 400 gpurun_out/bench_r04c_bench.json
if [ -f gpurun_out/r04cp/failed.txt ]; then cat gpurun_out/r04cp/failed.txt; fi
