# A/B on one box: the RPN head's 3x3 conv (and the box head's FC layers) on the single-accumulator form.  Edits the box's scratch copy only.
out=gpurun_out/oneacc; mkdir -p $out
run() { timeout 300 python bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 > $out/$1.json
  python - $out/$1.json $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p = d["timed_batch_parity"]
print(sys.argv[2], d["value"], d["ms_per_step"], p["gate"], p["deviation_among_matched"], p["bars"])
PY
}
run A1
sed -i 's/self.conv.two_acc = True/self.conv.two_acc = False/' lvc_amd/modeling/proposal_generator/rpn.py
run B1
sed -i 's/def pack_linear(weight, bias=None, split=None, two_acc=True)/def pack_linear(weight, bias=None, split=None, two_acc=False)/' lvc_amd/kernels.py
run C1
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_e2e.py tests/test_gpu_golden.py -q 2>&1 | tail -15 > $out/pytest_C.txt; tail -5 $out/pytest_C.txt
git checkout lvc_amd/kernels.py 2>/dev/null || sed -i 's/def pack_linear(weight, bias=None, split=None, two_acc=False)/def pack_linear(weight, bias=None, split=None, two_acc=True)/' lvc_amd/kernels.py
run B2
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_e2e.py -q 2>&1 | tail -15 > $out/pytest_B.txt; tail -5 $out/pytest_B.txt
sed -i 's/self.conv.two_acc = False/self.conv.two_acc = True/' lvc_amd/modeling/proposal_generator/rpn.py
run A2
