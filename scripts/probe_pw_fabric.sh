# Fabric traffic (FETCH_SIZE x 2 per the gfx950 correction, WRITE_SIZE) and TCC hit / miss of single pointwise launches: is the operand ingest
# of the pipelined pointwise kernel served by L2 or does it cross the fabric?  usage: bash scripts/probe_pw_fabric.sh   (one rocprofv3 --pmc pass per counter set)
out=$GRAFT_REPO_ROOT/gpurun_out/pwfabric; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for shape in "1 12544 8000 1 1024 1 1 0" "8 1024 50 84 256 1 1 0" "8 256 50 84 1024 1 1 0" "8 256 200 336 256 1 1 0"; do
  tag=$(echo $shape | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
    ct=$(echo $c | tr ' ' '+')
    timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_$ct -o p -- python $GRAFT_REPO_ROOT/scripts/probe_one.py $shape > /dev/null 2>&1
  done
done
python - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pwfabric"
res = collections.defaultdict(dict)
for f in glob.glob(out + "/*/p_counter_collection.csv"):
    tag = os.path.basename(os.path.dirname(f))
    shape = tag.split("_FETCH")[0].split("_WRITE")[0].split("_TCC")[0].split("_TCP")[0]
    rows = [r for r in csv.DictReader(open(f)) if "conv_pw" in r["Kernel_Name"] or "pw_s1" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows: by[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    last = {}
    for (cn, d), v in by.items(): last.setdefault(cn, {})[int(d)] = sum(v)
    for cn, dd in last.items(): res[shape][cn] = dd[max(dd)]
for shape, d in sorted(res.items()):
    print(shape, {k: round(v, 1) for k, v in d.items()})
PY
