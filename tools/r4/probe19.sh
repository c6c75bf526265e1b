#!/bin/bash
# round 4, probe 19: is test_ln_fold_every_consumer_epilogue_and_tile flaky, and since when?  (whole kernel-test file, no -x)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
: > $O/r04p19.txt
for rep in 1 2 3 4 5; do
  for arm in new prev; do
    if [ $arm = prev ]; then d=$GRAFT_REPO_ROOT/tools/_alt/prev; else d=$GRAFT_REPO_ROOT; fi
    r=$(cd $d && timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tr '\n' ' ')
    echo "$arm rep$rep: $r" | tee -a $O/r04p19.txt
  done
done
