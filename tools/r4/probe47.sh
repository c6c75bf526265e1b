cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 800 python -m pytest tests/test_gpu_11_bench.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
