#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/e2e_ab.py --workload configs1 --arms default n1024=16 n1024=17 n1024=22 --rounds 3 --steps 5 > $O/r04p23_e2e_configs1.txt 2>&1
tail -5 $O/r04p23_e2e_configs1.txt
