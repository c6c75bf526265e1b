# K-loop ablations of the lock-step 128 x 128 tile, COMPILE-TIME (tools/_alt/abl<bits>: -DLEMAS_PHASE_TIMESTAMPS -DLEMAS_ABLATE=<bits>; 1 = no MFMAs,
# 2 = no refill LDS-DMA, 4 = no fragment reads): launch time and the in-kernel loop / epilogue stamps
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
{
for shape in "gemm_gate 1920 1024 1024" "gemm_gate 1920 1024 2048" "gemm_gelu 1920 2048 1024"; do
  for tile in 17 26; do
    for abl in 0 1 2 4 3 7; do
      (cd $GRAFT_REPO_ROOT/tools/_alt/abl$abl && python tools/kbench.py one $shape $tile --iters 50 2>&1 | grep -v "amdgpu.ids" | sed "s/^/ablate=$abl  /" | sed 's/span [0-9.]* us | first round: start +[0-9.]* //; s/ | later.*//')
    done
  done
done
} > $O/r04p44_kloop_ablations_compile_time.txt 2>&1
cat $O/r04p44_kloop_ablations_compile_time.txt
