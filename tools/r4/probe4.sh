set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_03_vocos.py tests/test_gpu_01_kernels.py tests/test_gpu_04_prosody.py tests/test_mel_frontend.py tests/test_uvr5_shell.py tests/test_gpu_12_graph_cache.py -x -q -p no:cacheprovider -m gpu 2>&1 | tail -6 > $O/r04p4_tests.txt
timeout 900 python bench.py --no-cpu-baseline --no-clock-power --steps 4 > $O/r04p4_bench.json 2> $O/r04p4_bench.err
cat $O/r04p4_tests.txt
python - <<PY
import json
d=json.load(open("$O/r04p4_bench.json"))
print(d["value"], d["phase_ms"], d["roofline_vocoder"], d["kernel_avg_us"])
PY
