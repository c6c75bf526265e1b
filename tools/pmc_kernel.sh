# Per-kernel PMC breakdown of ONE micro-benchmarked kernel (development aid): separate rocprofv3 --pmc passes, one counter group each.
#   gpurun -- 'bash tools/pmc_kernel.sh attention 1875 16 0 0 attn_fwd  > gpurun_out/pmc_attn.txt'
# args: <what> <M> <N> <K> <tile> <kernel-name-substring>
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
what=$1; M=$2; N=$3; K=$4; tile=$5; sub=$6
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  d=/tmp/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  (cd $R && rocprofv3 --kernel-trace --pmc $grp -d $d -- python tools/kbench.py one $what $M $N $K $tile --iters 5 > /dev/null 2>/tmp/pmc_err.log) || tail -3 /tmp/pmc_err.log
  (cd $R && python tools/rocpd_pmc.py $(find $d -name "*_results.db" | head -1) $sub)
done
