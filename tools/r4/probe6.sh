set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python tools/r4/fp8_outlier_points.py configs0_outlier_nfe32 > $O/r04p6_fp8_sites.txt 2>&1
python tools/e2e_ab.py --workload configs3 --arms default gx=4 gx=2 n1024=16 n1024=17 n2048=16 --rounds 2 --steps 3 > $O/r04p6_e2e_configs3.txt 2>&1
cat $O/r04p6_fp8_sites.txt $O/r04p6_e2e_configs3.txt
