mkdir -p gpurun_out/r3x
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r3x/pytest.txt
timeout 600 python bench.py > gpurun_out/r3x/bench.json 2> gpurun_out/r3x/bench.err
for m in 128 48; do LVC_HALO_H2_MIN_TILES=$m timeout 200 python bench.py --no-extras --no-cpu-baseline --no-live-pmc --pipeline-depth 1 --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MIN_TILES $m', d['value'], d['ms_per_step'])" >> gpurun_out/r3x/min_tiles.txt; done
timeout 200 python scripts/probe_vit.py 64 > gpurun_out/r3x/vit.txt 2>&1
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3x/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-live-pmc --pipeline-depth 1 > $GRAFT_REPO_ROOT/gpurun_out/r3x/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/r3x/pytest.txt; cat gpurun_out/r3x/min_tiles.txt; ls gpurun_out/r3x/prof | head
