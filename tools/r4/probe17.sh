#!/bin/bash
# round 4, probe 17: with the cheaper epilogues, do the tile choices / the ln fold still stand?  (one engine, interleaved arms)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/e2e_ab.py --workload configs1 --arms default ln_fold=1 n1024=18 n1024=26 n2048=17 n2048=16 qkv=17 qkv=26 --rounds 3 --steps 5 > $O/r04p17_e2e_configs1.txt 2>&1
timeout 900 python tools/e2e_ab.py --workload configs3 --arms default ln_fold=1 n1024=16 n2048=16 n1024=17 --rounds 2 --steps 3 > $O/r04p17_e2e_configs3.txt 2>&1
timeout 600 python tools/e2e_ab.py --workload short --arms default ln_fold=1 n1024=19 n2048=18 --rounds 3 --steps 5 > $O/r04p17_e2e_short.txt 2>&1
tail -12 $O/r04p17_e2e_configs1.txt; tail -8 $O/r04p17_e2e_configs3.txt; tail -7 $O/r04p17_e2e_short.txt
