#!/bin/bash
# Round-2 evidence run (through gpurun): one rocprofv3 profile per tracking variant on the BASELINE config it serves,
# then the full bench.py line.  Summaries: python tools/save_profile_variant.py <tag> for every tag below.
set -o pipefail
mkdir -p gpurun_out/r02p
SPECS=${PAGK_PROFILE_SPECS:-"r02_a_cfg1:1:1000:0 r02_b_cfg4:4:4000:2 r02_c_cfg3:3:20000:3 r02_d_cfg3:3:20000:5 r02_d_cfg2x:2:2000:0 r02_e_cfg3:3:20000:6"}
for spec in $SPECS; do
  spec=${spec//:/ }
  set -- $spec
  timeout -k 10 240 bash tools/profile_variant.sh $1 $2 $3 $4 > gpurun_out/r02p/$1.log 2>&1 || echo "profile $1 failed" >> gpurun_out/r02p/failed.txt
  echo "profiled $1"
done
[ -n "$PAGK_PROFILE_NO_BENCH" ] && exit 0
timeout -k 10 400 python3 bench.py > gpurun_out/r02p/bench.json 2> gpurun_out/r02p/bench.err || echo "bench failed" >> gpurun_out/r02p/failed.txt
tail -c 600 gpurun_out/r02p/bench.err
head -c 1500 gpurun_out/r02p/bench.json
