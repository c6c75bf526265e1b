set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
out=$O/r04p13_pmc_diag.txt; rm -f $out
try() { label=$1; shift; d=/tmp/pmcd_$label; rm -rf $d
  (cd $R && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d -- python -X faulthandler bench.py --no-cpu-baseline --no-clock-power --steps 1 --warmup 1 --workload configs3 "$@" > /tmp/pmcd.out 2>/tmp/pmcd.log)
  echo "### $label ($*): rc $?" >> $out
  grep -A7 "Fatal Python" /tmp/pmcd.log | cut -c1-160 >> $out; }
try nograph --graph 0 --vocoder-graph 0
try depth8 --depth 8
try c2depth22 --workload configs2
cat $out
