# Round-4 probe 1 (baseline tree): graph capture cost, vendor-library GEMM ceiling, kernel micro-benchmarks, in-situ PMC groups.
#   gpurun --timeout 1500 -- 'bash tools/r4/probe1.sh'
set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python tools/graph_cost.py > $O/r04_graph_cost_baseline.txt 2>&1
python tools/exp/library_gemm_ceiling.py > $O/r04_library_gemm_ceiling.txt 2>&1
python tools/kbench.py gemm --M 1920 9216 18432 --tiles 0 16 17 22 26 > $O/r04_kbench_gemm_baseline.txt 2>&1
python tools/kbench.py attn --N 1875 1125 --BH 16 32 128 256 > $O/r04_kbench_attn_baseline.txt 2>&1
cd /tmp
rm -f $O/r04_pmc_groups_baseline.txt
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  d=/tmp/pmcg_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  (cd $R && timeout 400 rocprofv3 --kernel-trace --pmc $grp -d $d -- python bench.py --no-cpu-baseline --no-clock-power --steps 1 --warmup 1 > /tmp/pmcg.out 2>/tmp/pmcg.log) || tail -5 /tmp/pmcg.log
  echo "## $grp" >> $O/r04_pmc_groups_baseline.txt
  (cd $R && python tools/rocpd_pmc.py $(find $d -name "*_results.db" | head -1) gemm_bf16 gemm_pp attn_fwd ln_mod gemm_qkv >> $O/r04_pmc_groups_baseline.txt)
done
cat $O/r04_graph_cost_baseline.txt; cat $O/r04_library_gemm_ceiling.txt; cat $O/r04_kbench_gemm_baseline.txt; cat $O/r04_kbench_attn_baseline.txt; cat $O/r04_pmc_groups_baseline.txt
