#!/bin/bash
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/gaps
rocprofv3 --kernel-trace --output-format csv -d $OUT -o g -- python $GRAFT_REPO_ROOT/scripts/probe_gaps.py > /dev/null 2>&1
python - <<PY
import csv, statistics as st
rows = sorted(csv.DictReader(open("$OUT/g_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "").replace("conv3x3_halo_h2_kernel", "halo").replace("conv_pw_dma_kernel", "pw")
gaps = {}
for a, b in zip(rows, rows[1:]):
    key = short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])
    gaps.setdefault(key, []).append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for k, v in sorted(gaps.items()):
    if len(v) >= 10:
        print("%-40s n %3d  median gap %6.2f us  min %6.2f" % (k, len(v), st.median(v), min(v)))
PY
