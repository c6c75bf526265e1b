#!/bin/bash
# round 4, probe 27: convpos loads batched: tests + one-lease A/B against commit 6cb18d5
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_01_kernels.py tests/test_gpu_02_fp8.py tests/test_gpu_00_sample.py -x -q -m gpu > $O/r04p27_tests.txt 2>&1
tail -5 $O/r04p27_tests.txt
: > $O/r04p27_ab.txt
for w in configs1 configs3 short; do
  for rep in 1 2 3; do
    for arm in prev new; do
      if [ $arm = prev ]; then d=$GRAFT_REPO_ROOT/tools/_alt/prev; else d=$GRAFT_REPO_ROOT; fi
      v=$(cd $d && timeout 600 python bench.py --workload $w --no-phases 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))")
      echo "$w rep$rep $arm $v" | tee -a $O/r04p27_ab.txt
    done
  done
done
