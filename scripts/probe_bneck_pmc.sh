#!/bin/bash
# Counters of conv_bneck_kernel (identity block, 8 x 200 x 336): rocprofv3 --pmc passes (kernel-trace only), one small set per pass
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_bneck; mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p$i -- python $GRAFT_REPO_ROOT/scripts/probe_bneck.py time > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_bneck"
agg = {}; n = {}
for f in sorted(glob.glob(d + "/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "conv_bneck" in r["Kernel_Name"]:
            key = (r["Kernel_Name"][5:36], r["Counter_Name"])
            agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"]); n[key] = n.get(key, 0) + 1
for k in sorted(agg): print(k[0], k[1], "%.5g per launch" % (agg[k] / n[k]))
PY
