cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/r4/skip_dead_tail.py configs2_prosody_b8 > $O/r04p35_skip_dead_tail.txt 2>&1
timeout 300 python tools/r4/skip_dead_tail.py mini_batch >> $O/r04p35_skip_dead_tail.txt 2>&1
cat $O/r04p35_skip_dead_tail.txt | tail -12
