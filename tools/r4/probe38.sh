cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for sd in 0 1 0 1; do
timeout 600 python bench.py --workload configs2 --no-cpu-baseline --no-clock-power --no-phases --steps 4 --warmup 1 --skip-dead $sd > $O/r04p38_configs2_skip$sd.json 2> $O/r04p38_configs2_skip$sd.err || tail -5 $O/r04p38_configs2_skip$sd.err
python - <<PY
import json
d=json.load(open("$O/r04p38_configs2_skip$sd.json"))
print("skip_dead=$sd", d["value"], d["ms_per_step"], d["mel_mse_vs_reference"], d["config"]["rows_computed_per_step"], d["config"]["padded_row_waste"], d["config"]["padded_row_waste_at_batch_pitch"])
PY
done
timeout 600 python tools/r4/skip_dead_tail.py configs2_prosody_b8 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/e2e_ab.py --workload configs1 --arms default --rounds 2 --steps 4 2>&1 | tail -1
timeout 600 python tools/e2e_ab.py --workload configs3 --arms default --rounds 2 --steps 3 2>&1 | tail -1
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/r04p38_tests.txt
cat $O/r04p38_tests.txt
