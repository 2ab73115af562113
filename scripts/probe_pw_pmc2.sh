#!/bin/bash
# Counters of conv_pw_s1_kernel on res4's two pointwise shapes (8 x 50 x 84): conv1 1024 -> 256 and conv3 256 -> 1024
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_pw2; mkdir -p $out
for shape in "1024 256" "256 1024"; do
  set -- $shape
  i=0
  for cset in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
              "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
              "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
              "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVES"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $cset --kernel-trace --output-format csv -d $out -o c$1_p$i -- python $GRAFT_REPO_ROOT/scripts/probe_one.py 8 $1 50 84 $2 1 1 0 > /dev/null 2>&1
  done
done
python - <<'PY'
import csv, glob, os
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_pw2"
for c in ("1024", "256"):
    agg = {}; n = {}
    for f in sorted(glob.glob(d + "/c%s_p*_counter_collection.csv" % c)):
        for r in csv.DictReader(open(f)):
            if "conv_pw" in r["Kernel_Name"]:
                key = (r["Kernel_Name"][5:36], r["Counter_Name"])
                agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"]); n[key] = n.get(key, 0) + 1
    print("== input channels", c)
    for k in sorted(agg): print(k[0], k[1], "%.5g" % (agg[k] / n[k]))
PY
