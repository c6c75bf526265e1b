# Round-4 probe 3: full GPU suite on the new tree (ABI 200, graph cache, vocoder graph, new f32 GEMM), every bench workload, the sharded job
set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python tools/r4/fp8_outlier_points.py > $O/r04p3_fp8_outlier_points.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -30 > $O/r04p3_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 >> $O/r04p3_tests.txt
for w in configs1 configs2 configs3 configs4 short; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline > $O/r04p3_bench_$w.json 2> $O/r04p3_bench_$w.err || tail -5 $O/r04p3_bench_$w.err
done
timeout 900 python bench.py --job configs3_full --steps 2 --warmup 1 > $O/r04p3_job_n1.json 2> $O/r04p3_job_n1.err || tail -5 $O/r04p3_job_n1.err
cat $O/r04p3_fp8_outlier_points.txt $O/r04p3_tests.txt
for w in configs1 configs2 configs3 configs4 short; do cut -c1-330 $O/r04p3_bench_$w.json; python - <<PY
import json
d=json.load(open("$O/r04p3_bench_$w.json"))
print({k:d.get(k) for k in ("value","ms_per_step","mel_mse_vs_reference","phase_ms")}, d["roofline_vocoder"], d["kernel_avg_us"])
PY
done
cut -c1-900 $O/r04p3_job_n1.json
