#!/bin/bash
# rocprofv3 of ONE tracking variant on ONE workload (via tools/run_once.py): kernel trace + separate --pmc passes.
# Usage (through gpurun):  bash tools/profile_variant.sh <tag> <config> <n> <kernel>     -> gpurun_out/prof_<tag>/
# Then: python tools/save_profile_variant.py <tag>   copies kernel_stats.csv + pmc_summary.json into profiles/<tag>/.
set -o pipefail
TAG=$1; CFG=$2; N=$3; KERN=$4
export TMPDIR=/tmp
# the package's process-level runtime default (runtime_env.py) -- exported here because the profiler's preloaded library initialises HIP before python starts
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=${DEBUG_CLR_GRAPH_PACKET_CAPTURE:-0}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
run() { python3 $REPO/tools/run_once.py $CFG $N $KERN; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/run_once.py $CFG $N $KERN 20 > $OUT/trace.log 2>&1 || exit 1
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/tools/run_once.py $CFG $N $KERN 6 > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/failed.txt; }
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVE_CYCLES
pass pmc_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
pass pmc_sq3 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_VALU_MFMA_BUSY_CYCLES
# TA/TCP/TCC derived counters abort rocprofv3 on this image (unknown names); not collected


ls $OUT
