# The 3x3 kernel's patch shape forced for every launch (LVC_HALO_PATCH=PH,PW: an experiment hook of csrc/conv3x3_halo_s1.hip, compiled in only by `make -C lvc_amd/csrc HOOKS=-DLVC_HALO_PATCH_HOOK` after a `make clean`) against the chooser's own: bench value per shape.
for p in default 16,16 14,18 12,21 10,25 8,32 18,14 21,12; do
  if [ $p = default ]; then unset LVC_HALO_PATCH; else export LVC_HALO_PATCH=$p; fi
  python bench.py --steps 30 --warmup 6 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
unset LVC_HALO_PATCH
python bench.py --steps 30 --warmup 6 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default again', d['value'], d['ms_per_step'], d['roofline']['frac'])"
