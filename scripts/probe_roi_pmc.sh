#!/bin/bash
# ROIAlign of a real step: time and fabric reads per launch with / without the XCD-local row mapping (LVC_ROI_XCD_ROWS)
cd /tmp; export TMPDIR=/tmp
for x in 1 0; do
  export LVC_ROI_XCD_ROWS=$x
  OUT=$GRAFT_REPO_ROOT/gpurun_out/roi_pmc/x$x
  python $GRAFT_REPO_ROOT/scripts/probe_roi.py 2>&1 | tail -1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o t1 -- python $GRAFT_REPO_ROOT/scripts/probe_roi.py > /dev/null 2>&1
  python - <<PY
import csv, glob
tot = n = 0
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "roi_align_fwd_nhwc_lds" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print("  LVC_ROI_XCD_ROWS=$x: FETCH_SIZE x 2 per launch = %.3f GB (%d launches)" % (tot * 1024 * 2 / 1e9 / max(1, n), n))
PY
done
