mkdir -p gpurun_out/r02c
./tools/exp/memset_race_probe > gpurun_out/r02c/memset_probe.txt 2>&1
cat gpurun_out/r02c/memset_probe.txt
python -c "
import torch, time
a = torch.randn(4096, 4096, device='cuda'); b = torch.randn(4096, 4096, device='cuda')
t0 = time.time()
while time.time() - t0 < 420:
    for _ in range(20): c = a @ b
    torch.cuda.synchronize(); time.sleep(0.002)
" &
BG=$!
timeout 200 python tools/repro_config3.py --reps 6 --modes eager,default,default,default,nograph,nodual > gpurun_out/r02c/repro_bgload.txt 2>&1
tail -n 2 gpurun_out/r02c/repro_bgload.txt
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r02c/full_pytest_bgload.txt 2>&1
tail -n 5 gpurun_out/r02c/full_pytest_bgload.txt
kill $BG 2>/dev/null; wait $BG 2>/dev/null
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r02c/full_pytest.txt 2>&1
tail -n 5 gpurun_out/r02c/full_pytest.txt
