#!/bin/bash
# LDS conflicts and instruction mix of the dominant 3x3 kernel (p2 shape); LVC_HALO_PATCH is passed through
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o h -- python $GRAFT_REPO_ROOT/scripts/probe_one.py 8 256 200 336 256 3 1 1 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_lds"
agg = {}; name = ""
for f in sorted(glob.glob(d + "/h_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "halo" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = float(r["Counter_Value"]); name = r["Kernel_Name"][:40]
print(os.environ.get("LVC_HALO_PATCH", "default patch"), name, {k: "%.4g" % v for k, v in agg.items()})
PY
