# Round-4 closing evidence on one MI355X box (every step under its own timeout): full GPU suite, smoke, every bench line (five workloads, configs2
# with the padding blocks skipped, fp8, the sharded job, the N > 1 rehearsals), graph cost, rocprofv3 kernel stats -> gpurun_out/<TAG>_*
# (the PMC passes of evidence3.sh are not repeated: the block-chain kernels' loops and epilogues are those of r04e)
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'TAG=r04h bash tools/r4/evidence4.sh'      then copy what is to be judged into profiles/
set -x
TAG=${TAG:-r04h}
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/${TAG}_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 >> $O/${TAG}_tests.txt
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
for w in configs2 configs3 configs4 short; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/${TAG}_bench_$w.json 2>> $O/${TAG}_bench.err
done
timeout 600 python bench.py --workload configs2 --skip-dead 1 --no-cpu-baseline > $O/${TAG}_bench_configs2_skip_dead.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --fp8 1 --no-cpu-baseline > $O/${TAG}_bench_fp8.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --workload configs4 --fp8 0 --no-cpu-baseline > $O/${TAG}_bench_configs4_bf16.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --job configs3_full --steps 2 --warmup 1 > $O/${TAG}_job_configs3_full_n1.json 2>> $O/${TAG}_bench.err
LEMAS_SHARE_GPU=1 LEMAS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --no-cpu-baseline --steps 4 > $O/${TAG}_bench_2ranks_shared_gpu.json 2>> $O/${TAG}_bench.err
LEMAS_FORCE_DIST=1 LEMAS_DIST_BACKEND=nccl timeout 600 python bench.py --no-cpu-baseline --steps 4 > $O/${TAG}_bench_rccl_world1.json 2>> $O/${TAG}_bench.err
cd /tmp
for w in configs1 configs2 configs3; do
  rm -rf /tmp/prof_$w
  x=""; [ $w = configs2 ] && x="--skip-dead 1"
  (cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -- python bench.py --workload $w $x --no-cpu-baseline --no-clock-power --steps 4 > /tmp/prof_$w.out 2>/tmp/prof_$w.log)
  (cd $R && python tools/rocpd_summary.py $(find /tmp/prof_$w -name "*_results.db" | head -1) > $O/${TAG}_kernel_stats_$w.txt)
  tail -1 /tmp/prof_$w.out > $O/${TAG}_bench_under_rocprof_$w.json
done
set +x
cat $O/${TAG}_tests.txt; cut -c1-400 $O/${TAG}_bench.json
for f in configs2 configs2_skip_dead configs3 configs4 short fp8 configs4_bf16 2ranks_shared_gpu rccl_world1; do cut -c1-170 $O/${TAG}_bench_$f.json; done
cut -c1-170 $O/${TAG}_job_configs3_full_n1.json
head -12 $O/${TAG}_kernel_stats_configs1.txt | cut -c1-70,110-175
tail -5 $O/${TAG}_bench.err
