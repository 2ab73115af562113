#!/bin/bash
# clock / time / MFMA-busy of ablated halo-kernel builds (build/ablate/lib_h<id>.so) on the p2 3x3 shape
cd /tmp; export TMPDIR=/tmp
SO=$GRAFT_REPO_ROOT/lvc_amd/liblvc_amd.so
cp $SO $SO.orig
for a in "$@"; do
  cp $GRAFT_REPO_ROOT/build/ablate/lib_h$a.so $SO
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_habl -o a$a -- python $GRAFT_REPO_ROOT/scripts/probe_one.py 8 256 200 336 256 3 1 1 > /dev/null 2>&1
done
cp $SO.orig $SO
