set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider -k "not bench and not configs" 2>&1 | tail -6 > $O/r04p7_tests.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_voc_$c
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_voc_$c -- python tools/r4/vocos_only.py 938 10 > /tmp/voc_$c.out 2>/tmp/voc_$c.log) || tail -3 /tmp/voc_$c.log
done
python tools/r4/vocoder_traffic.py $(find /tmp/pmc_voc_FETCH_SIZE -name "*_results.db" | head -1) $(find /tmp/pmc_voc_WRITE_SIZE -name "*_results.db" | head -1) 13 > $O/r04p7_vocoder_traffic.json
python tools/r4/vocos_only.py 938 50 > $O/r04p7_vocoder_alone.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-clock-power --steps 6 > $O/r04p7_bench.json 2> $O/r04p7_bench.err
cat $O/r04p7_tests.txt $O/r04p7_vocoder_traffic.json $O/r04p7_vocoder_alone.txt
python - <<PY
import json
d=json.load(open("$O/r04p7_bench.json"))
print(d["value"], d["phase_ms"], d["roofline_vocoder"], d["roofline"]["limited_by"], d["roofline"]["traffic"], d["roofline"]["frac_rocprof"])
PY
