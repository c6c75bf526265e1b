cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for sd in 1 0 1 0; do
timeout 600 python bench.py --workload configs2 --no-cpu-baseline --no-clock-power --no-phases --steps 4 --warmup 1 --skip-dead $sd > $O/r04p34_configs2_skip$sd.json 2> $O/r04p34_configs2_skip$sd.err || tail -5 $O/r04p34_configs2_skip$sd.err
python - <<PY
import json
d=json.load(open("$O/r04p34_configs2_skip$sd.json"))
print("skip_dead=$sd", d["value"], d["ms_per_step"], d["mel_mse_vs_reference"], d["config"]["rows_computed_per_step"], d["config"]["padded_row_waste"])
PY
done
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider -k "not bench" 2>&1 | tail -8 > $O/r04p34_tests.txt
cat $O/r04p34_tests.txt
