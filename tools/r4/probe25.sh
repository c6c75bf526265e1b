#!/bin/bash
# round 4, probe 25: fp32 GEMM with loads that actually overlap the MFMAs (zeroing at the park, two register sets): parity + vocoder / hoist times
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_01_kernels.py tests/test_gpu_03_vocos.py tests/test_gpu_04_prosody.py tests/test_gpu_00_sample.py -x -q -m gpu > $O/r04p25_tests.txt 2>&1; tail -3 $O/r04p25_tests.txt
for arm in prev new; do
  if [ $arm = prev ]; then d=$GRAFT_REPO_ROOT/tools/_alt/prev; else d=$GRAFT_REPO_ROOT; fi
  echo "== $arm"; (cd $d && timeout 300 python tools/r4/vocos_only.py 938 50 2>&1 | grep -v amdgpu | tail -3; timeout 300 python tools/r4/vocos_only.py 2813 20 2>&1 | grep -v amdgpu | tail -2)
  (cd $d && timeout 600 python bench.py --workload configs1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('configs1', round(d['value'],2), d['phase_ms'])")
done | tee $O/r04p25_vocoder.txt
