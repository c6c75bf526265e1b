#!/bin/bash
# round 4, probe 15: one-lease A/B of the row-window epilogues: the previous commit's tree (tools/_alt/prev, its own library) against this tree,
# interleaved, bench.py lines only
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/r04p15_ab.txt
for w in configs1 configs3 short; do
  for rep in 1 2 3; do
    for arm in prev new; do
      if [ $arm = prev ]; then d=$GRAFT_REPO_ROOT/tools/_alt/prev; else d=$GRAFT_REPO_ROOT; fi
      v=$(cd $d && timeout 600 python bench.py --workload $w --no-phases 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), d.get('clock_power',{}).get('sclk_mhz'))")
      echo "$w rep$rep $arm $v" | tee -a $O/r04p15_ab.txt
    done
  done
done
