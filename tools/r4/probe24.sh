#!/bin/bash
# round 4, probe 24: residual rows of the next block prefetched in the 256 x 256 tile's gate + residual epilogue (56 B of scratch): kernel bench + configs3 / configs4 A/B
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "gemm" > $O/r04p24_tests.txt 2>&1; tail -2 $O/r04p24_tests.txt
for arm in prev new; do
  if [ $arm = prev ]; then d=$GRAFT_REPO_ROOT/tools/_alt/prev; else d=$GRAFT_REPO_ROOT; fi
  echo "== $arm"; (cd $d && timeout 600 python tools/kbench.py gemm --M 9216 18432 30720 --tiles 22 2>&1 | grep -v amdgpu)
done | tee $O/r04p24_kbench.txt
: > $O/r04p24_ab.txt
for w in configs3 configs2; do
  for rep in 1 2 3; do
    for arm in prev new; do
      if [ $arm = prev ]; then d=$GRAFT_REPO_ROOT/tools/_alt/prev; else d=$GRAFT_REPO_ROOT; fi
      v=$(cd $d && timeout 600 python bench.py --workload $w --no-phases 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))")
      echo "$w rep$rep $arm $v" | tee -a $O/r04p24_ab.txt
    done
  done
done
