#!/bin/bash
# MFMA-pipe duty and clock of the f16x2 256-row pointwise kernel on a compute-heavy GEMM
cd /tmp; export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm -o g1 -- python $GRAFT_REPO_ROOT/scripts/probe_gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm -o g2 -- python $GRAFT_REPO_ROOT/scripts/probe_gemm_one.py "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_gemm"
agg = {}
for f in sorted(glob.glob(d + "/g*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "pw256" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = float(r["Counter_Value"])
ms = None
for f in sorted(glob.glob(d + "/g1*kernel_trace.csv")):
    dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "pw256" in r["Kernel_Name"]]
    ms = dd[-1]
g = agg.get("GRBM_GUI_ACTIVE", 0) / 8
print("ms %.3f  clock %.2f GHz  mfma_busy %.1f%%" % (ms, g / ms / 1e6, 100 * agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024)))
print({k: "%.4g" % v for k, v in agg.items()})
PY
