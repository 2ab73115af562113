# Same-box A/B of a module constant: bash scripts/ab_env.sh <module> <CONSTANT> [rounds] -- alternates bench.py with the constant as committed / set False.
mod=$1; name=$2; rounds=${3:-2}; out=gpurun_out/ab_$name; mkdir -p $out
for i in $(seq 1 $rounds); do
  for v in True False; do
    timeout 300 python - $mod $name $v <<'PY' 2>/dev/null | tail -1 > $out/${v}_$i.json
import importlib, runpy, sys
m = importlib.import_module(sys.argv[1]); setattr(m, sys.argv[2], sys.argv[3] == "True")
sys.argv = ["bench.py", "--steps", "40", "--warmup", "8", "--no-extras", "--no-cpu-baseline", "--no-live-pmc"]
runpy.run_path("bench.py", run_name="__main__")
PY
    python -c "
import json,sys
d=json.loads(open('$out/${v}_$i.json').read().strip().splitlines()[-1]); print('$name=$v', d['value'], d['ms_per_step'], d['value_inference_batched']['value'])"
  done
done
