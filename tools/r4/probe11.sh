# (1) XCD-aligned blocks vs round 3's equal runs, end to end on one engine; (2) does the profiler's abort on the batched workloads follow the tile order?
set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python tools/e2e_ab.py --workload configs1 --arms default runs=1 default runs=1 --rounds 3 --steps 5 > $O/r04p11_e2e_xcd_configs1.txt 2>&1
python tools/e2e_ab.py --workload configs3 --arms default runs=1 --rounds 3 --steps 3 > $O/r04p11_e2e_xcd_configs3.txt 2>&1
out=$O/r04p11_pmc_diag.txt; rm -f $out
cd /tmp
d=/tmp/pmcd_runs; rm -rf $d
(cd $R && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d -- python -X faulthandler bench.py --no-cpu-baseline --no-clock-power --steps 1 --warmup 1 --workload configs3 --xcd-runs 1 > /tmp/pmcd.out 2>/tmp/pmcd.log)
echo "### configs3 --xcd-runs 1: rc $?" >> $out
grep -A8 "Fatal Python" /tmp/pmcd.log | cut -c1-200 >> $out
db=$(find $d -name "*_results.db" | head -1)
[ -n "$db" ] && (cd $R && python tools/rocpd_pmc.py $db gemm_pp attn_fwd 2>&1 | head -6 | cut -c1-120 >> $out)
cat $O/r04p11_e2e_xcd_configs1.txt $O/r04p11_e2e_xcd_configs3.txt $out
