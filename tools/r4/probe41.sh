cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_07_api.py tests/test_gpu_05_edges.py -x -q -m gpu -p no:cacheprovider -s -k "batched_lines or ragged" 2>&1 | grep -v amdgpu.ids | tail -12
