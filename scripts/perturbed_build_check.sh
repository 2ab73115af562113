#!/bin/bash
# VERDICT r4 #3 "done" criterion: a deliberately perturbed build must fail the accuracy gates.  Variants (bash scripts/build_variant.sh ...):
#   build/variants/dropcross.so     conv3x3_wino.hip -DWN_DIAG_DROP_CROSS: the activation-residual x weight product (one of the three MFMAs per
#                                   fp32-accurate product) removed from conv3x3_wino_kernel -- 8 of the full-size step's 60 conv launches (not reached
#                                   by smoke's small images)
#   build/variants/dropcross_pw.so  conv_pw_s1.hip -DPW_DIAG_DROP_CROSS: the same term removed from every pointwise / FC layer
#   python scripts/perturb_run.py post_topk950 ...  (round 6) a HOST-side perturbation with the shipped library: 950 instead of 1000 proposals kept per image
#                                   (~5 % of the proposals dropped): the identity bars are measured (oracle/noise.py), a hand-set 90 % let this pass
# Expected: smoke() raises, the bench's timed_batch_parity gate fails (exit 4), the e2e tests fail; the shipped library passes the same three.
# usage (on the GPU box): bash scripts/perturbed_build_check.sh > gpurun_out/perturbed.txt
for lib in "" build/variants/dropcross.so build/variants/dropcross_pw.so; do
  echo "=== LVC_AMD_LIB=${lib:-<the shipped library>}"
  export LVC_AMD_LIB=$lib
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300; echo "smoke rc=${PIPESTATUS[0]}"
  python bench.py --steps 5 --warmup 2 --no-extras --no-live-pmc > /tmp/b.json 2> /tmp/b.err; echo "bench rc=$?"; tail -2 /tmp/b.err | cut -c1-300
  python -m pytest tests/test_gpu_e2e.py -q -x 2>&1 | tail -2
done
echo "=== perturbation post_topk950 (the shipped library; scripts/perturb_run.py)"
unset LVC_AMD_LIB
python scripts/perturb_run.py post_topk950 smoke 2>&1 | tail -1 | cut -c1-300; echo "smoke rc=${PIPESTATUS[0]}"
python scripts/perturb_run.py post_topk950 bench.py --steps 5 --warmup 2 --no-extras --no-live-pmc > /tmp/b.json 2> /tmp/b.err; echo "bench rc=$?"; tail -2 /tmp/b.err | cut -c1-300
python scripts/perturb_run.py post_topk950 pytest tests/test_gpu_e2e.py -q -x 2>&1 | tail -2
