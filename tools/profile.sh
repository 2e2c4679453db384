#!/bin/bash
# rocprofv3 runs of bench.py on the GPU box; summaries land in gpurun_out/prof_* (copy the ones to
# keep into profiles/).  Usage (via gpurun): bash tools/profile.sh <tag>
set -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
# the package's process-level runtime default (runtime_env.py) -- exported here because the profiler's preloaded library initialises HIP before python starts
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=${DEBUG_CLR_GRAPH_PACKET_CAPTURE:-0}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
# 1. kernel trace + stats (per-kernel time)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench_trace.json 2> $OUT/trace.log || exit 1
# 2. HBM traffic counters, separate passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.log || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.log || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.log || exit 1
find $OUT -name "*.csv" | head -20
