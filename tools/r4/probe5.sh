set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider -s -k "fp8 or kernels or sample or configs or bench or abi" 2>&1 | grep -v "^$" | tail -40 > $O/r04p5_tests.txt
for w in configs1 configs3; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline --no-clock-power --steps 6 > $O/r04p5_bench_$w.json 2> $O/r04p5_bench_$w.err || tail -5 $O/r04p5_bench_$w.err
done
cd /tmp
rm -f $O/r04p5_pmc.txt
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pmcg_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  (cd $R && timeout 400 rocprofv3 --kernel-trace --pmc $grp -d $d -- python bench.py --no-cpu-baseline --no-clock-power --steps 1 --warmup 1 > /tmp/pmcg.out 2>/tmp/pmcg.log) || tail -5 /tmp/pmcg.log
  echo "## $grp" >> $O/r04p5_pmc.txt
  (cd $R && python tools/rocpd_pmc.py $(find $d -name "*_results.db" | head -1) gemm_bf16 gemm_pp attn_fwd ln_mod gemm_qkv >> $O/r04p5_pmc.txt)
done
cat $O/r04p5_tests.txt; cut -c1-200 $O/r04p5_bench_configs1.json; cut -c1-200 $O/r04p5_bench_configs3.json; cat $O/r04p5_pmc.txt
