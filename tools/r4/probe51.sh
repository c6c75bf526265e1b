cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_07_api.py tests/test_gpu_11_bench.py -x -q -m gpu -p no:cacheprovider -s -k "batched_lines or spawns_its_own" 2>&1 | grep -v amdgpu.ids | tail -6
