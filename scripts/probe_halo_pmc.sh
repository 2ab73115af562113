#!/bin/bash
# PMC passes over one 3x3 shape for the halo kernel (LVC_CONV_HALO=1) and the generic kernel (=0)
cd /tmp; export TMPDIR=/tmp
for h in 1 0; do
for ctr in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  n=$(echo $ctr | tr ' ' '_')
  LVC_CONV_HALO=$h rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_halo -o h${h}_$n -- python $GRAFT_REPO_ROOT/scripts/probe_one.py 8 256 200 336 256 3 1 1 > /dev/null 2>&1
done; done
