cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT/tools/_alt/phases
{
for shape in "gemm_gate 1920 1024 1024" "gemm_gate 1920 1024 2048"; do
  for abl in 0 7 1 2 4; do
    python tools/kbench.py one $shape $((17 + 256 * abl)) --iters 50 2>&1 | grep -v "amdgpu.ids" | sed "s/^/ablate=$abl  /"
  done
done
} > $O/r04p43_kloop_ablation_phases.txt 2>&1
cat $O/r04p43_kloop_ablation_phases.txt | cut -c1-260
