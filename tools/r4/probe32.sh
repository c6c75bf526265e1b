cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
run() { what=$1; M=$2; N=$3; K=$4; tile=$5; sub=$6
  echo "== $what M=$M N=$N K=$K tile=$tile"
  for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" "TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TA_BUSY_avr" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
    d=/tmp/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)
    rm -rf $d
    (cd $R && timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $d -- python tools/kbench.py one $what $M $N $K $tile --iters 5 > /dev/null 2>/tmp/pmc_err.log) || tail -3 /tmp/pmc_err.log
    (cd $R && python tools/rocpd_pmc.py $(find $d -name "*_results.db" | head -1) $sub)
  done
}
{
run gemm_gate 1920 1024 1024 17 gemm_bf16
run gemm_gelu 1920 2048 1024 26 gemm_bf16
run gemm_gelu 9216 2048 1024 22 gemm_pp
} > $O/r04p32_pmc_vmem_path.txt 2>&1
cat $O/r04p32_pmc_vmem_path.txt
