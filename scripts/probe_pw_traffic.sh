#!/bin/bash
# fabric traffic per launch of the non-3x3 layers (scripts/probe_layers_list.py runs each 11 times in order)
cd /tmp; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_pwt -o $ctr -- python $GRAFT_REPO_ROOT/scripts/probe_layers_list.py > /dev/null 2>&1
done
