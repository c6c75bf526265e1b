cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|Counter_Name)?.*\b(TA_|TCP_|TD_|GRBM_)" | sed 's/Description.*//' | awk '{$1=$1};1' | sort -u | head -150 > $O/r04p31_counters.txt
wc -l $O/r04p31_counters.txt
head -100 $O/r04p31_counters.txt
