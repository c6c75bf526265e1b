# the same rocprofv3 --pmc pass over configs[3] on the ROUND-3 tree (tools/_alt/r03: git worktree of 256ecd7, built in place): does the
# profiler abort there too?
set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
out=$O/r04p9_pmc_diag_r03_tree.txt; rm -f $out
d=/tmp/pmcd_r03; rm -rf $d
(cd $R/tools/_alt/r03 && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d -- python bench.py --no-cpu-baseline --no-clock-power --steps 1 --warmup 1 --workload configs3 > /tmp/pmcd.out 2>/tmp/pmcd.log)
echo "### round-3 tree, configs3, FETCH_SIZE: rc $?" >> $out
db=$(find $d -name "*_results.db" | head -1)
[ -n "$db" ] && (cd $R && python tools/rocpd_pmc.py $db gemm_pp attn_fwd 2>&1 | head -6 | cut -c1-120 >> $out)
tail -4 /tmp/pmcd.log | cut -c1-300 >> $out
cat $out
