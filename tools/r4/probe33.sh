cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/e2e_ab.py --workload configs1 --arms default n2048=22 qkv_fused=0 qkv_fused=0,n2048=22 n1024=16 --rounds 3 --steps 4 > $O/r04p33_e2e_configs1.txt 2>&1
tail -6 $O/r04p33_e2e_configs1.txt
