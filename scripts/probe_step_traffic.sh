#!/bin/bash
# Fabric traffic of ONE bench step per kernel launch (rocprofv3 --pmc FETCH_SIZE WRITE_SIZE, kernel-trace only), listed by launch in step order
# with the kernel's grid: which launches move more than their layer's input + output + weights.
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/step_traffic; mkdir -p $out
# one counter per pass (MI355X_MICROARCH.md; both in one pass hung the run for its whole time limit)
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o f -- python $GRAFT_REPO_ROOT/scripts/probe_step_pmc.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out -o w -- python $GRAFT_REPO_ROOT/scripts/probe_step_pmc.py > /dev/null 2>&1
python - <<'PY'
import csv, os, collections
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/step_traffic"
by = collections.OrderedDict()
for tag in ("f", "w"):
    for r in csv.DictReader(open(d + "/%s_counter_collection.csv" % tag)):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        by.setdefault(k, {})[r["Counter_Name"]] = by.setdefault(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
items = sorted(by.items())
marks = [i for i, (k, v) in enumerate(items) if "gelu" in k[1]]
a, b = marks[0], marks[1]
tot = 0.0
agg = collections.OrderedDict()
for (disp, name, grid), v in items[a + 1:b]:
    gb = (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 / 1e9
    tot += gb
    print("%-60s grid %-8s fetch %7.3f GB (x2 corrected) write %7.3f GB" % (name, grid, 2 * v.get("FETCH_SIZE", 0) * 1024 / 1e9, v.get("WRITE_SIZE", 0) * 1024 / 1e9))
    e = agg.setdefault(name.split("(")[0], [0, 0.0]); e[0] += 1; e[1] += gb
print("step total %.2f GB" % tot)
for k, (n, gb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-60s n %3d  %7.3f GB" % (k, n, gb))
PY
