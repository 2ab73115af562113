# Same-box A/B of an environment switch: bash scripts/ab_envvar.sh NAME VALUE [rounds] -- alternates bench.py without / with NAME=VALUE.
name=$1; val=$2; rounds=${3:-2}; out=gpurun_out/ab_$name; mkdir -p $out
for i in $(seq 1 $rounds); do
  for v in unset set; do
    if [ $v = set ]; then export $name=$val; else unset $name; fi
    timeout 300 python bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 > $out/${v}_$i.json
    python -c "
import json
d=json.loads(open('$out/${v}_$i.json').read().strip().splitlines()[-1]); print('$name $v', d['value'], d['ms_per_step'], d['value_inference_batched']['value'])"
  done
done
unset $name
