# One gpurun call at the end of a round: the GPU test suite, smoke(), the full default bench line, and the rocprofv3 kernel trace of the
# same command (summaries are copied into profiles/ by hand afterwards).  usage: gpurun --timeout 2400 -- 'bash scripts/round_end_run.sh r4x'
tag=${1:-r5x}
mkdir -p gpurun_out/$tag
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/$tag/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.txt 2>&1
timeout 600 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-live-pmc --pipeline-depth 1 > $GRAFT_REPO_ROOT/gpurun_out/$tag/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/$tag/pytest.txt; tail -2 gpurun_out/$tag/smoke.txt; python - <<PY
import json
d = json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], d["ms_per_step"], "batched", d["value_inference_batched"]["value"], "pipelined", d["pipelined"]["value"], "graphed", (d.get("graphed") or {}).get("value"))
print({k: r.get(k) for k in ("frac", "achieved", "traffic", "backbone_mfma_busy", "step_traffic", "step_traffic_over_algorithmic")})
print("parity", d["timed_batch_parity"]["gate"], d["timed_batch_parity"]["deviation_among_matched"]["box_median"], d["timed_batch_parity"]["bars"]["box_median"])
print("knn", d["knn"].get("ms_per_sweep"), {k: v.get("ms") for k, v in d["bandwidth_kernels"].items() if isinstance(v, dict)})
print("cpu", d["cpu_baseline"]["value"])
PY
