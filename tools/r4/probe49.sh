cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_05_edges.py tests/test_gpu_07_api.py -x -q -m gpu -p no:cacheprovider -k "ragged or batched_lines" 2>&1 | tail -4
timeout 600 python tools/r4/skip_dead_tail.py configs2_prosody_b8 2>&1 | grep -v amdgpu.ids | tee $O/r04p49_skip_modes_tail.txt
for rep in 1 2; do
for mode in "0 0" "1 0" "1 1" "1 2"; do
set -- $mode
timeout 600 python bench.py --workload configs2 --no-cpu-baseline --no-clock-power --no-phases --steps 4 --warmup 1 --skip-masked $1 --skip-dead $2 > $O/r04p49_configs2_m$1_d$2.json 2> $O/r04p49.err || tail -5 $O/r04p49.err
python - <<PY
import json
d=json.load(open("$O/r04p49_configs2_m$1_d$2.json"))
print("skip_masked=$1 skip_dead=$2", round(d["value"],2), round(d["ms_per_step"],1), d["mel_mse_vs_reference"], d["config"]["rows_computed_attention_half"], d["config"]["rows_computed_per_step"])
PY
done
done 2>&1 | tee $O/r04p49_configs2_modes.txt
