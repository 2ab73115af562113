#!/bin/bash
# Counters of conv3x3_wino_kernel on the FPN-output p2 shape (8 x 200 x 336, 256 -> 256): two rocprofv3 --pmc passes (kernel-trace only)
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_wino; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o a -- python $GRAFT_REPO_ROOT/scripts/probe_one.py 8 256 200 336 256 3 1 1 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out -o b -- python $GRAFT_REPO_ROOT/scripts/probe_one.py 8 256 200 336 256 3 1 1 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_wino"
for tag in ("a", "b"):
    agg = {}; n = {}
    for f in sorted(glob.glob(d + "/%s_counter_collection.csv" % tag)):
        for r in csv.DictReader(open(f)):
            if "conv3x3" in r["Kernel_Name"]:
                key = (r["Kernel_Name"][:28], r["Counter_Name"])
                agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"]); n[key] = n.get(key, 0) + 1
    for k in sorted(agg): print(k[0], k[1], "%.4g per launch" % (agg[k] / n[k]))
PY
