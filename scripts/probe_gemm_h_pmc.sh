#!/bin/bash
# fp16 pre-filter GEMM alone: time + fabric reads per launch for tile-order variants (LVC_GH_NGROUP)
cd /tmp; export TMPDIR=/tmp
for ng in 10 5 4 3 2; do
  export LVC_GH_NGROUP=$ng
  OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_h_pmc/ng$ng
  python $GRAFT_REPO_ROOT/scripts/probe_gemm_h.py 2>&1 | tail -1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o t1 -- python $GRAFT_REPO_ROOT/scripts/probe_gemm_h.py > /dev/null 2>&1
  python - <<PY
import csv, glob
tot = n = 0
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
launches = 23    # probe_gemm_h.py: 3 warm-up + 20 timed launches
print("  ngroup $ng: FETCH_SIZE x 2 per launch = %.3f GB (%d counter rows)" % (tot * 1024 * 2 / 1e9 / launches, n))
PY
done
