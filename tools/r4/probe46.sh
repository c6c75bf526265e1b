cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
# the driver's own launcher for N > 1, rehearsed with two ranks on the box's one GPU (gloo for the collectives: RCCL refuses two ranks on one device)
LEMAS_SHARE_GPU=1 LEMAS_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > $O/r04p46_torchrun_2ranks.out 2> $O/r04p46_torchrun_2ranks.err
echo "rc=$?"; grep -c '"metric"' $O/r04p46_torchrun_2ranks.out; tail -1 $O/r04p46_torchrun_2ranks.out | cut -c1-300; tail -3 $O/r04p46_torchrun_2ranks.err
