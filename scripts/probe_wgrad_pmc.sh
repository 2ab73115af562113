#!/bin/bash
# MFMA duty, LDS conflicts and instruction mix of the fp16 weight-gradient kernel
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_wgrad; mkdir -p $out
export WGRAD_SPLIT=${WGRAD_SPLIT-f16x2}   # WGRAD_SPLIT= (empty) profiles the default bf16x3 kernel
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out -o g1 -- python $GRAFT_REPO_ROOT/scripts/probe_wgrad_one.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o g2 -- python $GRAFT_REPO_ROOT/scripts/probe_wgrad_one.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out -o g3 -- python $GRAFT_REPO_ROOT/scripts/probe_wgrad_one.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_wgrad"
agg = {}
for f in sorted(glob.glob(d + "/g*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = float(r["Counter_Value"])
ms = None
for f in sorted(glob.glob(d + "/g1*kernel_trace.csv")):
    dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "wgrad" in r["Kernel_Name"]]
    ms = dd[-1]
g = agg.get("GRBM_GUI_ACTIVE", 0) / 8
print("ms %.3f  clock %.2f GHz  mfma_busy %.1f%%" % (ms, g / ms / 1e6, 100 * agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024)))
print({k: "%.4g" % v for k, v in agg.items()})
PY
