#!/bin/bash
# Round-4 evidence run (through gpurun): rocprofv3 profiles (kernel trace + separate --pmc passes) of the bench.py headline,
# of the pipelined 4-wave kernel on configs[1] / [2], of the level kernel on configs[3] and of the batched multi-camera
# launch; then the full bench.py line.  Summaries: python tools/save_profile.py r04_bench ; python tools/save_profile_variant.py <tag>.
set -o pipefail
mkdir -p gpurun_out/r04p
SPECS=${PAGK_PROFILE_SPECS:-"r04_a_cfg1:1:1000:0 r04_a_cfg2:2:2000:0 r04_f_cfg3:3:20000:7"}
for spec in $SPECS; do
  spec=${spec//:/ }
  set -- $spec
  timeout -k 10 240 bash tools/profile_variant.sh $1 $2 $3 $4 > gpurun_out/r04p/$1.log 2>&1 || echo "profile $1 failed" >> gpurun_out/r04p/failed.txt
  echo "profiled $1"
done
# the batched launch: the same passes over tools/run_batch_once.py
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r04_batch_cfg4x8
rm -rf $OUT; mkdir -p $OUT
( cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/run_batch_once.py 20 > $OUT/trace.log 2>&1 || echo "batch trace failed" >> $REPO/gpurun_out/r04p/failed.txt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/run_batch_once.py 6 > $OUT/pmc_fetch.log 2>&1 || echo "batch fetch failed" >> $REPO/gpurun_out/r04p/failed.txt
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/run_batch_once.py 6 > $OUT/pmc_write.log 2>&1 || echo "batch write failed" >> $REPO/gpurun_out/r04p/failed.txt
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq1 -- python3 $REPO/tools/run_batch_once.py 6 > $OUT/pmc_sq1.log 2>&1 || echo "batch sq failed" >> $REPO/gpurun_out/r04p/failed.txt )
echo "profiled batch"
timeout -k 10 300 bash tools/profile.sh r04_bench > gpurun_out/r04p/profile_bench.log 2>&1 || echo "profile bench failed" >> gpurun_out/r04p/failed.txt
echo "profiled bench"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_r04_bench.json 2> gpurun_out/r04p/bench.err || echo "bench failed" >> gpurun_out/r04p/failed.txt
tail -c 300 gpurun_out/r04p/bench.err
head -c 400 gpurun_out/bench_r04_bench.json
if [ -f gpurun_out/r04p/failed.txt ]; then cat gpurun_out/r04p/failed.txt; fi
