#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_cmp; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for k in 0 3; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/a$k -- python3 $REPO/tools/run_once.py 3 20000 $k > $OUT/a$k.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/b$k -- python3 $REPO/tools/run_once.py 3 20000 $k > $OUT/b$k.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/c$k -- python3 $REPO/tools/run_once.py 3 20000 $k > $OUT/c$k.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for k in (0, 3):
    vals = {}
    for part in "abc":
        for f in glob.glob(f"/root/repo/gpurun_out/pmc_cmp/{part}{k}/*/*_counter_collection.csv"):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "k_track_block" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c, v in agg.items():
                vals[c] = sum(v) / len(v)
    print("kernel", k, {c: f"{v:.4g}" for c, v in sorted(vals.items())})
PY
