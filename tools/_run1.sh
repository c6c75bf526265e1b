mkdir -p gpurun_out/r02a
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_abi_errors.py tests/test_gpu_api.py tests/test_gpu_c_client.py tests/test_gpu_cli.py tests/test_gpu_configs.py -q -m gpu -k "not config2 and not config5 and not config1" 2>&1 | tail -15 > gpurun_out/r02a/pytest_$i.txt
done
for i in 1 2 3; do
  timeout 300 python tools/repro_config3.py --reps 3 > gpurun_out/r02a/repro_$i.txt 2>&1
done
LEMAS_GEMM_WIDE=16 timeout 300 python tools/repro_config3.py --reps 3 > gpurun_out/r02a/repro_wide16.txt 2>&1
timeout 300 python tools/repro_config3.py --reps 2 --oracle --modes eager,default > gpurun_out/r02a/repro_oracle.txt 2>&1
tail -3 gpurun_out/r02a/*.txt
