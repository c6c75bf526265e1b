cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_00_sample.py tests/test_speech_edit.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
rm -rf /tmp/prof_c1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c1 -- python bench.py --no-cpu-baseline --no-clock-power --no-phases --steps 4 > /tmp/prof_c1.out 2>/tmp/prof_c1.log
python tools/rocpd_summary.py $(find /tmp/prof_c1 -name "*_results.db" | head -1) > $O/r04k_kernel_stats_configs1.txt
grep -c "at::native" $O/r04k_kernel_stats_configs1.txt; grep "at::native\|rocclr" $O/r04k_kernel_stats_configs1.txt | cut -c1-150
tail -1 /tmp/prof_c1.out | cut -c1-200
