# Round-4 evidence, part 2 (profiler passes with counters; one pass per counter group, each under its own timeout): PMC traffic of configs[3],
# the counter groups behind the per-kernel bounds for configs[1] and configs[3], the vocoder's bytes per decode
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'TAG=r04b bash tools/r4/evidence2.sh'
set -x
TAG=${TAG:-r04b}
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
pass() {   # pass <workload> <out file> <counters...>
  w=$1; out=$2; shift 2
  d=/tmp/pmcp_${w}_$(echo "$*" | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $d -- python bench.py --workload $w --no-cpu-baseline --no-clock-power --no-phases --steps 1 --warmup 1 > /tmp/pmcp.out 2>/tmp/pmcp.log) || { echo "# pass failed ($w: $*)" >> $out; tail -3 /tmp/pmcp.log >> $out; }
  echo "## $*" >> $out
  db=$(find $d -name "*_results.db" | head -1)
  [ -n "$db" ] && (cd $R && python tools/rocpd_pmc.py $db gemm_bf16 gemm_pp attn_fwd ln_mod gemm_qkv >> $out 2>&1)
}
for w in configs1 configs3; do
  rm -f $O/${TAG}_pmc_groups_$w.txt
  for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
    pass $w $O/${TAG}_pmc_groups_$w.txt $grp
  done
done
rm -f $O/${TAG}_pmc_configs3.txt
echo "### workload configs3" >> $O/${TAG}_pmc_configs3.txt
pass configs3 $O/${TAG}_pmc_configs3.txt FETCH_SIZE
pass configs3 $O/${TAG}_pmc_configs3.txt WRITE_SIZE
# the vocoder alone: bytes per decode
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_voc_$c
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_voc_$c -- python tools/r4/vocos_only.py 938 10 > /tmp/voc_$c.out 2>/tmp/voc_$c.log) || tail -3 /tmp/voc_$c.log
done
(cd $R && python tools/r4/vocoder_traffic.py $(find /tmp/pmc_voc_FETCH_SIZE -name "*_results.db" | head -1) $(find /tmp/pmc_voc_WRITE_SIZE -name "*_results.db" | head -1) 13 > $O/${TAG}_vocoder_traffic.json)
(cd $R && python tools/r4/vocos_only.py 938 50 > $O/${TAG}_vocoder_alone.txt 2>&1)
cat $O/${TAG}_vocoder_traffic.json $O/${TAG}_vocoder_alone.txt; head -40 $O/${TAG}_pmc_groups_configs3.txt | cut -c1-150; cat $O/${TAG}_pmc_configs3.txt | cut -c1-110
