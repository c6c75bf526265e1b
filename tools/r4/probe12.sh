set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
out=$O/r04p12_pmc_diag.txt; rm -f $out
d=/tmp/pmcd_np; rm -rf $d
(cd $R && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d -- python -X faulthandler bench.py --no-cpu-baseline --no-clock-power --no-phases --steps 1 --warmup 1 --workload configs3 > /tmp/pmcd.out 2>/tmp/pmcd.log)
echo "### configs3 --no-phases: rc $?" >> $out
grep -A8 "Fatal Python" /tmp/pmcd.log | cut -c1-200 >> $out
db=$(find $d -name "*_results.db" | head -1)
[ -n "$db" ] && (cd $R && python tools/rocpd_pmc.py $db gemm_pp attn_fwd 2>&1 | head -6 | cut -c1-120 >> $out)
cat $out
