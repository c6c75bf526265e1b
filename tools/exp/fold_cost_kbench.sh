for w in gemm_gate gemm_gate_fold; do for k in 1024 2048; do for t in 17 28; do python tools/kbench.py one $w 1920 1024 $k $t --iters 100; done; done; done
for w in gemm_gelu gemm_gelu_fold gemm_qk gemm_qk_fold; do for t in 26 16; do python tools/kbench.py one $w 1920 2048 1024 $t --iters 100; done; done
for w in gemm_v gemm_v_fold; do python tools/kbench.py one $w 1920 1024 1024 17 --iters 100; done
python tools/kbench.py one ln_mod 1920 1024 0 0 --iters 200
