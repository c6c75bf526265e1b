# K-loop ablations of the lock-step 128 x 128 tile on the measurement build (tools/_alt/phases: -DLEMAS_PHASE_TIMESTAMPS): what each pipe costs alone
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT/tools/_alt/phases
{
for shape in "gemm_gate 1920 1024 1024" "gemm_gate 1920 1024 2048" "gemm_gelu 1920 2048 1024"; do
  for tile in 17 26; do
    for abl in 0 1 2 4 3 5 6 7; do
      python tools/kbench.py one $shape $((tile + 256 * abl)) --iters 50 2>&1 | grep -v "amdgpu.ids\|phases" | sed "s/^/ablate=$abl  /"
    done
  done
done
} > $O/r04p42_kloop_ablations.txt 2>&1
cat $O/r04p42_kloop_ablations.txt
