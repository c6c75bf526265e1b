cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -p no:cacheprovider -k "every_tile_and_epilogue" 2>&1 | tail -3
timeout 300 python tools/kbench.py gemm --M 1920 --tiles 17 29 17 29 > $O/r04p45_kbench.txt 2>&1
cat $O/r04p45_kbench.txt
timeout 900 python tools/e2e_ab.py --workload configs1 --arms default n1024=29 --rounds 3 --steps 4 > $O/r04p45_e2e_configs1.txt 2>&1
tail -3 $O/r04p45_e2e_configs1.txt
