set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/r01o_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1 >> $O/r01o_tests.txt
python bench.py > $O/r01o_bench.json 2> $O/r01o_bench.err
python bench.py --fp8 1 --no-cpu-baseline > $O/r01o_bench_fp8.json 2>> $O/r01o_bench.err
cd /tmp
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_o -- python bench.py --no-cpu-baseline > /tmp/prof_o.out 2>/tmp/prof_o.log)
(cd $R && python tools/rocpd_summary.py $(find /tmp/prof_o -name "*_results.db" | head -1) > $O/r01o_kernel_stats.txt)
tail -1 /tmp/prof_o.out > $O/r01o_bench_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -- python bench.py --no-cpu-baseline --steps 1 --warmup 1 > /tmp/pmc_$c.out 2>/tmp/pmc_$c.log)
  echo "## $c" >> $O/r01o_pmc.txt
  (cd $R && python tools/rocpd_pmc.py $(find /tmp/pmc_$c -name "*_results.db" | head -1) gemm_bf16 attn_fwd ln_mod gemm_qkv >> $O/r01o_pmc.txt)
done
cat $O/r01o_tests.txt; cut -c1-330 $O/r01o_bench.json; cut -c1-120 $O/r01o_bench_fp8.json; head -12 $O/r01o_kernel_stats.txt | cut -c1-60,110-170; cat $O/r01o_pmc.txt
