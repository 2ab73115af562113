#!/bin/bash
# kNN sweep kernels: kernel trace + stats, then FETCH_SIZE / WRITE_SIZE / busy counters in separate --pmc passes (kernel trace only)
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/knn_pmc
KNN_INPUTS=${KNN_INPUTS:-randn}
CMD="python $GRAFT_REPO_ROOT/bench.py --workload knn --knn-inputs $KNN_INPUTS --steps 5 --warmup 2 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > /dev/null 2>&1
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT -o t$i -- $CMD > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, json
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/knn_pmc"
res = {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(t in k for t in ("gemm_f16", "knn_verify", "rownorm_h", "colmean")):
            continue
        d = res.setdefault(k, {})
        c = d.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
launches = {}
for f in glob.glob(out + "/**/stats_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0]
        if k in res:
            launches[k] = (int(r["Calls"]), float(r["AverageNs"]))
summary = {}
for k, d in res.items():
    calls, avg_ns = launches.get(k, (0, 0.0))
    per = {c: v[0] / max(1, v[1]) for c, v in d.items()}      # per (launch x counter row); rows = dimensions of a counter per launch
    rows = {c: v[1] for c, v in d.items()}
    tot = {c: v[0] for c, v in d.items()}
    n = max(1, calls)
    fetch = tot.get("FETCH_SIZE", 0.0) / n * 1024 * 2          # KB -> B; gfx950: wide coalesced reads counted at half (guide, HBM section)
    write = tot.get("WRITE_SIZE", 0.0) / n * 1024
    g = tot.get("GRBM_GUI_ACTIVE", 0.0) / n / 8
    summary[k] = {"launches_in_stats_pass": calls, "avg_us": round(avg_ns / 1e3, 1), "hbm_read_bytes_per_launch": round(fetch), "hbm_write_bytes_per_launch": round(write),
                  "mfma_busy_fraction": round(tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n / max(1.0, g * 1024), 4) if g else None,
                  "clock_GHz": round(g / (avg_ns), 3) if avg_ns else None,
                  "lds_bank_conflict_cycles_per_launch": round(tot.get("SQ_LDS_BANK_CONFLICT", 0.0) / n)}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
