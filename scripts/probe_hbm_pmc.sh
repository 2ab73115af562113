#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE in separate passes, as the microarchitecture guide prescribes) and duration of
# the bandwidth-bound kernels north_star names: ROIAlign, NMS, kNN top-k / normalisation.
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_hbm; mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o $ctr -- python $GRAFT_REPO_ROOT/scripts/probe_step_one.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, json, os
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_hbm"
want = ["roi_align_fwd", "nms_mask", "nms_reduce", "nms_prep", "knn_topk_vote", "rownorm_kernel", "det_row_stats", "rpn_topk_phase", "stem_pool", "conv_pw_bf16x3"]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    last = {}
    for r in csv.DictReader(open(d + "/%s_counter_collection.csv" % ctr)):
        for w in want:
            if w in r["Kernel_Name"]:
                last[w] = float(r["Counter_Value"])          # last launch of that kernel
    for w, v in last.items():
        res.setdefault(w, {})[ctr + "_KB_raw"] = v
dur = {}
for r in csv.DictReader(open(d + "/FETCH_SIZE_kernel_trace.csv")):
    for w in want:
        if w in r["Kernel_Name"]:
            dur[w] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for w in res:
    f, wr = res[w].get("FETCH_SIZE_KB_raw", 0.0), res[w].get("WRITE_SIZE_KB_raw", 0.0)
    res[w]["launch_us_under_pmc"] = dur.get(w)
    res[w]["bytes_fetch_x1"] = f * 1024
    res[w]["bytes_write"] = wr * 1024
    if dur.get(w):
        res[w]["GBps_fetch_x1_plus_write"] = round((f + wr) * 1024 / (dur[w] * 1e-6) / 1e9, 1)
        res[w]["GBps_fetch_x2_plus_write"] = round((2 * f + wr) * 1024 / (dur[w] * 1e-6) / 1e9, 1)
print(json.dumps(res, indent=1))
json.dump(res, open(d + "/summary.json", "w"), indent=1)
PY
