# Round-4 probe 2: q64 attention kernel (parity, micro-benchmark, end-to-end A/B), step-graph cache, ABI 200
set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -x -q -k "attention" -p no:cacheprovider 2>&1 | tail -15 > $O/r04p2_tests_attn.txt
timeout 900 python -m pytest tests/test_gpu_12_graph_cache.py tests/test_gpu_00_sample.py tests/test_gpu_10_c_client.py tests/test_gpu_09_abi_errors.py -x -q -s -p no:cacheprovider 2>&1 | tail -25 > $O/r04p2_tests_misc.txt
python tools/kbench.py attn --N 1875 --BH 16 32 --variants 19 4112 4113 4114 4115 > $O/r04p2_kbench_attn.txt 2>&1
python tools/kbench.py attn --N 1125 --BH 128 --variants 19 4112 4113 4114 4115 >> $O/r04p2_kbench_attn.txt 2>&1
python tools/kbench.py attn --N 2814 750 --BH 16 --variants 19 4112 4113 4114 4115 >> $O/r04p2_kbench_attn.txt 2>&1
python tools/e2e_ab.py --workload configs1 --arms default attn=4112 attn=4113 attn=4114 attn=4115 --rounds 3 --steps 5 > $O/r04p2_e2e_attn_configs1.txt 2>&1
python tools/e2e_ab.py --workload configs3 --arms default attn=4112 attn=4113 attn=4115 --rounds 2 --steps 3 > $O/r04p2_e2e_attn_configs3.txt 2>&1
python tools/e2e_ab.py --workload short --arms default attn=4112 attn=4113 --rounds 2 --steps 5 > $O/r04p2_e2e_attn_short.txt 2>&1
python tools/graph_cost.py > $O/r04p2_graph_cost.txt 2>&1
cat $O/r04p2_tests_attn.txt $O/r04p2_tests_misc.txt $O/r04p2_kbench_attn.txt $O/r04p2_e2e_attn_configs1.txt $O/r04p2_e2e_attn_configs3.txt $O/r04p2_e2e_attn_short.txt $O/r04p2_graph_cost.txt
