#!/bin/bash
SO=$GRAFT_REPO_ROOT/lvc_amd/liblvc_amd.so
cp $SO $SO.orig
for a in "$@"; do
  cp $GRAFT_REPO_ROOT/build/ablate/lib_roi$a.so $SO
  echo "ROI_LDS_MAX_PIX $a"; python $GRAFT_REPO_ROOT/scripts/probe_misc.py 2>&1 | grep -i "roi" | head -3
done
cp $SO.orig $SO
