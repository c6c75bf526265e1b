# Round-4 final evidence on one MI355X box (every step under its own timeout): full GPU suite, smoke, every bench line (five workloads, fp8, the
# sharded job, the N > 1 rehearsals), graph cost, rocprofv3 kernel stats per workload, PMC traffic (configs1, configs3, vocoder) and the PMC counter
# groups behind the per-kernel bounds (profiler passes with --no-phases: see DESIGN.md section 8, profiler note) -> gpurun_out/<TAG>_*
#   /usr/local/graft/bin/gpurun --timeout 4200 -- 'TAG=r04e bash tools/r4/evidence3.sh'      then copy what is to be judged into profiles/
set -x
TAG=${TAG:-r04e}
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/${TAG}_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 >> $O/${TAG}_tests.txt
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
for w in configs2 configs3 configs4 short; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/${TAG}_bench_$w.json 2>> $O/${TAG}_bench.err
done
timeout 600 python bench.py --fp8 1 --no-cpu-baseline > $O/${TAG}_bench_fp8.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --workload configs4 --fp8 0 --no-cpu-baseline > $O/${TAG}_bench_configs4_bf16.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --job configs3_full --steps 2 --warmup 1 > $O/${TAG}_job_configs3_full_n1.json 2>> $O/${TAG}_bench.err
LEMAS_SHARE_GPU=1 LEMAS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --job configs3_full --steps 1 --warmup 1 > $O/${TAG}_job_configs3_full_2ranks_shared_gpu.json 2>> $O/${TAG}_bench.err
LEMAS_SHARE_GPU=1 LEMAS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --no-cpu-baseline --steps 4 > $O/${TAG}_bench_2ranks_shared_gpu.json 2>> $O/${TAG}_bench.err
LEMAS_FORCE_DIST=1 LEMAS_DIST_BACKEND=nccl timeout 600 python bench.py --no-cpu-baseline --steps 4 > $O/${TAG}_bench_rccl_world1.json 2>> $O/${TAG}_bench.err
timeout 600 python tools/graph_cost.py > $O/${TAG}_graph_cost.txt 2>&1
cd /tmp
for w in configs1 configs2 configs3 configs4; do
  rm -rf /tmp/prof_$w
  (cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -- python bench.py --workload $w --no-cpu-baseline --no-clock-power --steps 4 > /tmp/prof_$w.out 2>/tmp/prof_$w.log)
  (cd $R && python tools/rocpd_summary.py $(find /tmp/prof_$w -name "*_results.db" | head -1) > $O/${TAG}_kernel_stats_$w.txt)
  tail -1 /tmp/prof_$w.out > $O/${TAG}_bench_under_rocprof_$w.json
done
pass() {   # pass <workload> <out file> <counters...>
  w=$1; out=$2; shift 2
  d=/tmp/pmcp_${w}_$(echo "$*" | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $d -- python bench.py --workload $w --no-cpu-baseline --no-clock-power --no-phases --steps 1 --warmup 1 > /tmp/pmcp.out 2>/tmp/pmcp.log) || { echo "# pass failed ($w: $*)" >> $out; tail -3 /tmp/pmcp.log >> $out; }
  echo "## $*" >> $out
  db=$(find $d -name "*_results.db" | head -1)
  [ -n "$db" ] && (cd $R && python tools/rocpd_pmc.py $db gemm_bf16 gemm_pp attn_fwd ln_mod gemm_qkv gemm_f32 convpos >> $out 2>&1)
}
rm -f $O/${TAG}_pmc.txt
for w in configs1 configs3; do
  echo "### workload $w" >> $O/${TAG}_pmc.txt
  pass $w $O/${TAG}_pmc.txt FETCH_SIZE
  pass $w $O/${TAG}_pmc.txt WRITE_SIZE
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_voc_$c
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_voc_$c -- python tools/r4/vocos_only.py 938 10 > /tmp/voc_$c.out 2>/tmp/voc_$c.log) || tail -3 /tmp/voc_$c.log
done
(cd $R && python tools/r4/vocoder_traffic.py $(find /tmp/pmc_voc_FETCH_SIZE -name "*_results.db" | head -1) $(find /tmp/pmc_voc_WRITE_SIZE -name "*_results.db" | head -1) 13 > $O/${TAG}_vocoder_traffic.json)
(cd $R && timeout 300 python tools/r4/vocos_only.py 938 50 > $O/${TAG}_vocoder_alone.txt 2>&1)
for w in configs1 configs3; do
  rm -f $O/${TAG}_pmc_groups_$w.txt
  for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
    pass $w $O/${TAG}_pmc_groups_$w.txt $grp
  done
done
set +x
cat $O/${TAG}_tests.txt; cut -c1-400 $O/${TAG}_bench.json
for f in configs2 configs3 configs4 short fp8 configs4_bf16 2ranks_shared_gpu rccl_world1; do cut -c1-170 $O/${TAG}_bench_$f.json; done
cut -c1-170 $O/${TAG}_job_configs3_full_n1.json; cut -c1-170 $O/${TAG}_job_configs3_full_2ranks_shared_gpu.json
cat $O/${TAG}_graph_cost.txt $O/${TAG}_vocoder_traffic.json $O/${TAG}_vocoder_alone.txt
head -14 $O/${TAG}_kernel_stats_configs1.txt | cut -c1-70,110-175
grep -c "pass failed" $O/${TAG}_pmc.txt $O/${TAG}_pmc_groups_configs1.txt $O/${TAG}_pmc_groups_configs3.txt
tail -5 $O/${TAG}_bench.err
