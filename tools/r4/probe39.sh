#!/bin/bash
# one-lease A/B of the dead-block plumbing on the DEFAULT path: the previous commit's tree (tools/_alt/prev) against this tree, interleaved
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/r04p39_ab.txt
for w in configs1 configs3; do
  for rep in 1 2 3; do
    for arm in prev new; do
      if [ $arm = prev ]; then d=$GRAFT_REPO_ROOT/tools/_alt/prev; else d=$GRAFT_REPO_ROOT; fi
      v=$(cd $d && timeout 600 python bench.py --workload $w --no-phases --no-cpu-baseline --no-clock-power --steps 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))")
      echo "$w rep$rep $arm $v" | tee -a $O/r04p39_ab.txt
    done
  done
done
