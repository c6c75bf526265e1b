# why do the rocprofv3 --pmc passes over the batched workloads abort on this tree?  one pass per suspect
set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
out=$O/r04p8_pmc_diag.txt; rm -f $out
try() {   # try <label> <bench args...>
  label=$1; shift
  d=/tmp/pmcd_$label; rm -rf $d
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d -- python bench.py --no-cpu-baseline --no-clock-power --steps 1 --warmup 1 "$@" > /tmp/pmcd.out 2>/tmp/pmcd.log)
  rc=$?
  echo "### $label ($*): rc $rc" >> $out
  grep -i -m5 "check failed\|fatal\|F0\|terminate\|assert\|error" /tmp/pmcd.log | cut -c1-400 >> $out
  db=$(find $d -name "*_results.db" | head -1)
  [ -n "$db" ] && (cd $R && python tools/rocpd_pmc.py $db gemm_pp attn_fwd 2>&1 | head -6 | cut -c1-120 >> $out)
}
try c3_default --workload configs3
try c3_novocgraph --workload configs3 --vocoder-graph 0
try c3_nograph --workload configs3 --graph 0
try c3_single_lane --workload configs3 --dual 0
try c3_depth2 --workload configs3 --depth 2
cat $out
