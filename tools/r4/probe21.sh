#!/bin/bash
# round 4, probe 21: the kernel-test file 30 times over (one rare failure of a ln-fold consumer case was seen in a suite run)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
: > $O/r04p21.txt
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 30); do
  r=$(timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|Error" | tr '\n' ' ')
  echo "rep$rep: $r" | tee -a $O/r04p21.txt
done
