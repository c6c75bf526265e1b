cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/r04j_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 >> $O/r04j_tests.txt
timeout 600 python bench.py --workload configs2 --no-cpu-baseline > $O/r04j_bench_configs2.json 2> $O/r04j_bench.err
timeout 600 python bench.py --workload configs2 --no-cpu-baseline --skip-dead 2 > $O/r04j_bench_configs2_skip_dead2.json 2>> $O/r04j_bench.err
timeout 600 python bench.py --workload configs2 --no-cpu-baseline --skip-dead 1 > $O/r04j_bench_configs2_skip_dead1.json 2>> $O/r04j_bench.err
timeout 600 python bench.py --workload configs2 --no-cpu-baseline --skip-masked 0 > $O/r04j_bench_configs2_skip_masked0.json 2>> $O/r04j_bench.err
cat $O/r04j_tests.txt
for f in configs2_skip_masked0 configs2 configs2_skip_dead2 configs2_skip_dead1; do python - <<PY
import json
d=json.load(open("$O/r04j_bench_$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],1), d["mel_mse_vs_reference"], d["config"]["rows_computed_attention_half"], d["config"]["rows_computed_per_step"])
PY
done
