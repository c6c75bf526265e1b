cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/r04i_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 >> $O/r04i_tests.txt
timeout 600 python bench.py > $O/r04i_bench.json 2> $O/r04i_bench.err
cat $O/r04i_tests.txt; cut -c1-300 $O/r04i_bench.json; python - <<PY
import json
d=json.load(open("$O/r04i_bench.json")); print(d["value"], d["roofline"].get("k_loop_pipes",{}).get("us_per_k_tile"), d["roofline_vocoder"]["frac"], d["cpu_baseline"]["value"])
PY
