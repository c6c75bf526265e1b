# Round evidence on one MI355X box: full GPU suite, smoke, bench lines, rocprofv3 kernel stats and PMC traffic -> gpurun_out/<TAG>_*
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'TAG=r03z bash tools/evidence.sh'      then copy what is to be judged into profiles/
set -x
TAG=${TAG:-r03}
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/${TAG}_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 >> $O/${TAG}_tests.txt
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --fp8 1 --no-cpu-baseline > $O/${TAG}_bench_fp8.json 2>> $O/${TAG}_bench.err
python bench.py --workload configs3 --no-cpu-baseline > $O/${TAG}_bench_configs3.json 2>> $O/${TAG}_bench.err
python bench.py --workload short --no-cpu-baseline > $O/${TAG}_bench_short.json 2>> $O/${TAG}_bench.err
LEMAS_SHARE_GPU=1 LEMAS_DIST_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline --steps 4 > $O/${TAG}_bench_2ranks_shared_gpu.json 2>> $O/${TAG}_bench.err
# every RCCL call of the multi-GPU path, in a world of one
LEMAS_FORCE_DIST=1 LEMAS_DIST_BACKEND=nccl python bench.py --no-cpu-baseline --steps 4 > $O/${TAG}_bench_rccl_world1.json 2>> $O/${TAG}_bench.err
cd /tmp
for w in configs1 configs3; do
  rm -rf /tmp/prof_$w
  (cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -- python bench.py --workload $w --no-cpu-baseline --no-clock-power > /tmp/prof_$w.out 2>/tmp/prof_$w.log)
  (cd $R && python tools/rocpd_summary.py $(find /tmp/prof_$w -name "*_results.db" | head -1) > $O/${TAG}_kernel_stats_$w.txt)
  tail -1 /tmp/prof_$w.out > $O/${TAG}_bench_under_rocprof_$w.json
done
rm -f $O/${TAG}_pmc.txt
for w in configs1 configs3; do
  echo "### workload $w" >> $O/${TAG}_pmc.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${w}_$c
    (cd $R && rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${w}_$c -- python bench.py --workload $w --no-cpu-baseline --no-clock-power --steps 1 --warmup 1 > /tmp/pmc_$c.out 2>/tmp/pmc_$c.log)
    echo "## $c" >> $O/${TAG}_pmc.txt
    (cd $R && python tools/rocpd_pmc.py $(find /tmp/pmc_${w}_$c -name "*_results.db" | head -1) gemm_bf16 gemm_pp attn_fwd ln_mod gemm_qkv gemm_f32 convpos >> $O/${TAG}_pmc.txt)
  done
done
cat $O/${TAG}_tests.txt; cut -c1-400 $O/${TAG}_bench.json; cut -c1-160 $O/${TAG}_bench_fp8.json; cut -c1-160 $O/${TAG}_bench_configs3.json; cut -c1-160 $O/${TAG}_bench_short.json; cut -c1-200 $O/${TAG}_bench_2ranks_shared_gpu.json; cut -c1-200 $O/${TAG}_bench_rccl_world1.json
head -14 $O/${TAG}_kernel_stats_configs1.txt | cut -c1-70,110-175; cat $O/${TAG}_pmc.txt
