#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
: > $O/r04p20.txt
for arm in new prev; do
  if [ $arm = prev ]; then d=$GRAFT_REPO_ROOT/tools/_alt/prev; else d=$GRAFT_REPO_ROOT; fi
  echo "== $arm" | tee -a $O/r04p20.txt
  (cd $d && timeout 900 python $GRAFT_REPO_ROOT/tools/r4/ln_fold_stress.py 150 2>&1 | grep -v amdgpu.ids | tee -a $O/r04p20.txt)
done
