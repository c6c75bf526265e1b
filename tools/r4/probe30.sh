set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python tools/e2e_ab.py --workload configs1 --arms default half=1 default half=1 --rounds 3 --steps 4 > $O/r04p30_e2e_configs1.txt 2>&1
tail -6 $O/r04p30_e2e_configs1.txt
timeout 900 python tools/e2e_ab.py --workload short --arms default half=1 --rounds 3 --steps 4 > $O/r04p30_e2e_short.txt 2>&1
tail -4 $O/r04p30_e2e_short.txt
