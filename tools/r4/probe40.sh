#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && timeout 900 python -m pytest tests/test_gpu_01_kernels.py tests/test_gpu_00_sample.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
: > $O/r04p40_ab.txt
for w in configs1 short configs3; do
  for rep in 1 2 3; do
    for arm in prev new; do
      if [ $arm = prev ]; then d=$GRAFT_REPO_ROOT/tools/_alt/prev; else d=$GRAFT_REPO_ROOT; fi
      v=$(cd $d && timeout 600 python bench.py --workload $w --no-phases --no-cpu-baseline --no-clock-power --steps 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), d['mel_mse_vs_reference'], d['kernel_avg_us'].get('ln_mod'))")
      echo "$w rep$rep $arm $v" | tee -a $O/r04p40_ab.txt
    done
  done
done
