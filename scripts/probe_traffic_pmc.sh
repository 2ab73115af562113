#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) + busy counters of the dominant launch shape on the halo kernel
cd /tmp; export TMPDIR=/tmp
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic -o t$i -- python $GRAFT_REPO_ROOT/scripts/probe_one.py 8 256 200 336 256 3 1 1 > /dev/null 2>&1
done
