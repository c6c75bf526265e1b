set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
out=$O/r04p10_pmc_diag_log.txt; rm -f $out
d=/tmp/pmcd_x; rm -rf $d
(cd $R && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d -- python -X faulthandler bench.py --no-cpu-baseline --no-clock-power --steps 1 --warmup 1 --workload configs3 > /tmp/pmcd.out 2>/tmp/pmcd.log)
echo "### rc $?" >> $out
grep -v "simple_timer\|amdgpu.ids" /tmp/pmcd.log | tail -80 | cut -c1-260 >> $out
d=/tmp/pmcd_y; rm -rf $d
(cd $R && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d -- python -X faulthandler bench.py --no-cpu-baseline --no-clock-power --steps 1 --warmup 1 --workload configs3 --overlap 0 > /tmp/pmcd.out 2>/tmp/pmcd2.log)
echo "### overlap 0: rc $?" >> $out
grep -v "simple_timer\|amdgpu.ids" /tmp/pmcd2.log | tail -30 | cut -c1-260 >> $out
cat $out
