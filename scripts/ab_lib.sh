# Same-box A/B of two library builds: bash scripts/ab_lib.sh <other.so> [rounds] -- alternates bench.py on lvc_amd/liblvc_amd.so and on the other library (LVC_AMD_LIB).
other=$1; rounds=${2:-2}; out=gpurun_out/ab_lib; mkdir -p $out
for i in $(seq 1 $rounds); do
  for v in new other; do
    if [ $v = other ]; then export LVC_AMD_LIB=$PWD/$other; else unset LVC_AMD_LIB; fi
    timeout 300 python bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 > $out/${v}_$i.json
    python -c "
import json
d=json.loads(open('$out/${v}_$i.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['value_inference_batched']['value'])"
  done
done
unset LVC_AMD_LIB
