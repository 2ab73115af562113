#!/bin/bash
# PMC passes over one pointwise layer (scripts/probe_pw_shape.py runs it; every rocprofv3 call under `timeout`): probe_pwdma_pmc.sh <tag> N H W C K stride res_mode
cd /tmp; export TMPDIR=/tmp
tag=$1; shift
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o p$i -- python $GRAFT_REPO_ROOT/scripts/probe_pw_shape.py "$@" > /dev/null 2>&1
done
python - <<PY
import csv, glob
agg = {}
ms = None
for f in sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "conv_pw" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = float(r["Counter_Value"])
for f in sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/p1*kernel_trace.csv")):
    dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "conv_pw" in r["Kernel_Name"]]
    ms = dd[-1]
g = agg.get("GRBM_GUI_ACTIVE", 0) / 8
print("$tag $@ : ms %.4f clock %.2f GHz" % (ms, g / ms / 1e6 if ms else 0))
wc = agg.get("SQ_WAVE_CYCLES", 1)
for k2 in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
    print("  %s / SQ_WAVE_CYCLES = %.3f" % (k2, agg.get(k2, 0) / wc))
print("  mfma busy %.1f %% of (cycles x 1024 SIMDs)" % (100 * agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1, g * 1024)))
print("  " + ", ".join("%s=%.4g" % kv for kv in sorted(agg.items())))
PY
