cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_05_edges.py -x -q -m gpu -p no:cacheprovider -k "ragged" 2>&1 | tail -15 > $O/r04p36_tests.txt
cat $O/r04p36_tests.txt
