#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_voc
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d /tmp/prof_voc -- python tools/r4/vocos_only.py 938 30 > /tmp/voc.out 2>/tmp/voc.log)
(cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py $(find /tmp/prof_voc -name "*_results.db" | head -1) > $O/r04p26_vocoder_kernels.txt)
head -30 $O/r04p26_vocoder_kernels.txt | cut -c1-60,100-180
