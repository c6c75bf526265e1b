#!/bin/bash
# round 4, probe 14: row-window epilogues (uniform sample / live-row window, buffer-descriptor stores): parity + kernel bench + end-to-end
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_01_kernels.py tests/test_gpu_02_fp8.py tests/test_gpu_00_sample.py tests/test_gpu_05_edges.py tests/test_gpu_06_configs.py tests/test_gpu_09_abi_errors.py -x -q -m gpu > $O/r04p14_tests.txt 2>&1
tail -5 $O/r04p14_tests.txt
timeout 600 python tools/kbench.py gemm --M 1920 9216 18432 --tiles 0 16 17 22 26 > $O/r04p14_kbench_gemm.txt 2>&1
cat $O/r04p14_kbench_gemm.txt
for w in configs1 configs3; do
  timeout 900 python bench.py --workload $w > $O/r04p14_bench_$w.txt 2>&1; tail -1 $O/r04p14_bench_$w.txt | cut -c1-400
done
