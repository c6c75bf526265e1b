#!/bin/bash
# ONE parametrised GPU session (replaces the per-probe scripts of rounds 3-4): a list of steps, each under its own timeout, all output under
# gpurun_out/<TAG>_*; copy what is to be judged into profiles/.
#
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'TAG=r05a bash tools/gpu_session.sh tests smoke bench rocprof:configs1 timeline'
#
# steps (any order, repeated as needed):
#   tests[:<pytest args>]        full `pytest -m gpu` suite, or e.g. tests:tests/test_gpu_02_fp8.py
#   smoke                        __graft_entry__.smoke()
#   bench[:<workload>[:<extra bench.py flags, '+' for spaces>]]     e.g. bench  bench:configs3  bench:configs1:--fp8+1
#   lines                        every bench line of the round's table (five workloads, fp8, bf16 configs4, the job, the N > 1 rehearsals)
#   ab:<workloads,comma>:<reps>:<key=v,key=v>[:<key=v,...>...]      tools/e2e_ab.py arms on ONE engine, interleaved
#   rocprof:<workload>           rocprofv3 --kernel-trace --stats over bench.py -> <TAG>_kernel_stats_<w>.txt
#                                (STEPGAPS=1 in the environment: also tools/rocpd_step_gaps.py -> <TAG>_step_gaps_<w>.txt)
#   pmc:<workload>               the PMC group passes (one group per pass, --no-phases) -> <TAG>_pmc_groups_<w>.txt
#   traffic:<workload>           FETCH_SIZE / WRITE_SIZE passes -> <TAG>_pmc_traffic_<w>.txt
#   timeline[:<workload>]        measurement build (-DLEMAS_PHASE_TIMESTAMPS) -> tools/timeline_step.py -> product build again
#   kbench[:<args, '+' for spaces>]   tools/kbench.py
#   mbuild / pbuild              rebuild the libraries with -DLEMAS_MEASUREMENT_BUILD (no phase stamps) / as the product
#   scale:<n>                    bench.py with n ranks sharing the one GPU over gloo (weak-scaling line + the sharded job)
#   py:<script, '+' for spaces>  python <script> (experiments under tools/exp or tools/)
#   pyprof:<name>:<script, '+' for spaces>   rocprofv3 --kernel-trace --stats over `python <script>` -> <TAG>_kernel_stats_<name>.txt
#                                (SEQ=<n> in the environment: also the last n dispatches in launch order -> <TAG>_sequence_<name>.txt)
#   pypmc:<name>:<script, '+' for spaces>    the PMC group passes + FETCH_SIZE / WRITE_SIZE over `python <script>` -> <TAG>_pmc_<name>.txt (all kernels)
set -u
TAG=${TAG:-r05}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p "$O"
cd "$R"
sp() { echo "$1" | tr '+' ' '; }
PMC_GROUPS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum")
KSUB="gemm_bf16 gemm_pp gemm_sk attn_fwd ln_mod gemm_qkv gemm_f32 convpos vocos"
for step in "$@"; do
  kind=${step%%:*}; rest=""; [ "$step" != "$kind" ] && rest=${step#*:}
  echo "=== $step" | tee -a "$O/${TAG}_session.log"
  case $kind in
    tests)
      args=${rest:-tests}; name=$(echo "$args" | tr '/ ' '__' | cut -c1-40)
      timeout 1800 python -m pytest $(sp "$args") -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > "$O/${TAG}_tests_${name}.txt"; tail -4 "$O/${TAG}_tests_${name}.txt" ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 | tee "$O/${TAG}_smoke.txt" ;;
    bench)
      w=${rest%%:*}; x=""; [ "$rest" != "$w" ] && x=$(sp "${rest#*:}"); w=${w:-configs1}
      f="$O/${TAG}_bench_${w}$(echo "$x" | tr -d ' -' | cut -c1-24).json"
      cb="--no-cpu-baseline"; [ "$w" = configs1 ] && [ -z "$x" ] && cb=""
      timeout 900 python bench.py --workload $w $x $cb > "$f" 2>> "$O/${TAG}_bench.err"; cut -c1-260 "$f" ;;
    lines)
      timeout 900 python bench.py > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err"; cut -c1-400 "$O/${TAG}_bench.json"
      for w in configs2 configs3 configs4 short; do
        timeout 600 python bench.py --workload $w --no-cpu-baseline > "$O/${TAG}_bench_$w.json" 2>> "$O/${TAG}_bench.err"; cut -c1-170 "$O/${TAG}_bench_$w.json"
      done
      timeout 600 python bench.py --workload configs2 --skip-dead 1 --no-cpu-baseline > "$O/${TAG}_bench_configs2_skip_dead.json" 2>> "$O/${TAG}_bench.err"
      timeout 600 python bench.py --fp8 1 --no-cpu-baseline > "$O/${TAG}_bench_fp8.json" 2>> "$O/${TAG}_bench.err"
      timeout 600 python bench.py --workload configs4 --fp8 0 --no-cpu-baseline > "$O/${TAG}_bench_configs4_bf16.json" 2>> "$O/${TAG}_bench.err"
      timeout 600 python bench.py --job configs3_full --steps 2 --warmup 1 > "$O/${TAG}_job_configs3_full_n1.json" 2>> "$O/${TAG}_bench.err"
      LEMAS_FORCE_DIST=1 LEMAS_DIST_BACKEND=nccl timeout 600 python bench.py --no-cpu-baseline --steps 4 > "$O/${TAG}_bench_rccl_world1.json" 2>> "$O/${TAG}_bench.err"
      for f in configs2_skip_dead fp8 configs4_bf16 rccl_world1; do cut -c1-170 "$O/${TAG}_bench_$f.json"; done; cut -c1-170 "$O/${TAG}_job_configs3_full_n1.json" ;;
    ab)
      IFS=: read -r ws reps arms <<< "$rest"
      for w in $(echo "$ws" | tr ',' ' '); do
        timeout 1500 python tools/e2e_ab.py --workload $w --rounds ${reps:-3} --arms $(echo "$arms" | tr ':' ' ') 2>&1 | grep -v amdgpu.ids | tee -a "$O/${TAG}_ab_$w.txt"
      done ;;
    rocprof)
      w=${rest:-configs1}; rm -rf /tmp/prof_$w
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -- python $R/bench.py --workload $w --no-cpu-baseline --no-clock-power --steps 4 > /tmp/prof_$w.out 2> /tmp/prof_$w.log)
      python tools/rocpd_summary.py "$(find /tmp/prof_$w -name '*_results.db' | head -1)" > "$O/${TAG}_kernel_stats_$w.txt"
      [ -n "${STEPGAPS:-}" ] && python tools/rocpd_step_gaps.py "$(find /tmp/prof_$w -name '*_results.db' | head -1)" > "$O/${TAG}_step_gaps_$w.txt"
      tail -1 /tmp/prof_$w.out > "$O/${TAG}_bench_under_rocprof_$w.json"; head -14 "$O/${TAG}_kernel_stats_$w.txt" | cut -c1-70,110-175 ;;
    pmc)
      w=${rest:-configs1}; : > "$O/${TAG}_pmc_groups_$w.txt"
      for grp in "${PMC_GROUPS[@]}"; do
        d=/tmp/pmc_${w}_$(echo $grp | tr ' ' '_' | cut -c1-40); rm -rf $d
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $d -- python $R/bench.py --workload $w --no-cpu-baseline --no-clock-power --no-phases --steps 1 --warmup 1 > /dev/null 2> /tmp/pmc_err.log) || tail -3 /tmp/pmc_err.log
        echo "## $grp" >> "$O/${TAG}_pmc_groups_$w.txt"
        python tools/rocpd_pmc.py "$(find $d -name '*_results.db' | head -1)" $KSUB >> "$O/${TAG}_pmc_groups_$w.txt"
      done; tail -12 "$O/${TAG}_pmc_groups_$w.txt" ;;
    traffic)
      w=${rest:-configs1}; echo "### workload $w" > "$O/${TAG}_pmc_traffic_$w.txt"      # (the header tools/traffic_json.py keys on)
      for c in FETCH_SIZE WRITE_SIZE; do
        d=/tmp/pmc_${w}_$c; rm -rf $d
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $d -- python $R/bench.py --workload $w --no-cpu-baseline --no-clock-power --no-phases --steps 1 --warmup 1 > /dev/null 2> /tmp/pmc_err.log) || tail -3 /tmp/pmc_err.log
        echo "## $c" >> "$O/${TAG}_pmc_traffic_$w.txt"
        python tools/rocpd_pmc.py "$(find $d -name '*_results.db' | head -1)" $KSUB >> "$O/${TAG}_pmc_traffic_$w.txt"
      done; tail -12 "$O/${TAG}_pmc_traffic_$w.txt" ;;
    timeline)
      LEMAS_EXTRA_HIPCC_FLAGS=-DLEMAS_PHASE_TIMESTAMPS timeout 900 python -c "from lemas_tts_amd import build; build.build_library(force=True)"
      for w in $(echo "${rest:-configs1}" | tr ',' ' '); do
        timeout 600 python tools/timeline_step.py --workload $w --json "$O/${TAG}_timeline_$w${TLOPT:+_$(echo $TLOPT | tr -d ' =')}.json" ${TLOPT:+--opt $TLOPT} 2>&1 | grep -v amdgpu.ids > "$O/${TAG}_timeline_step_$w${TLOPT:+_$(echo $TLOPT | tr -d ' =')}.txt"; head -40 "$O/${TAG}_timeline_step_$w${TLOPT:+_$(echo $TLOPT | tr -d ' =')}.txt"
      done
      timeout 900 python -c "from lemas_tts_amd import build; build.build_library(force=True)" ;;
    tbuild)     # measurement build WITH the phase stamps (-DLEMAS_PHASE_TIMESTAMPS: the ablation instantiations of attention.hip / gemm_bf16.hip); `pbuild` restores
      LEMAS_EXTRA_HIPCC_FLAGS=-DLEMAS_PHASE_TIMESTAMPS timeout 900 python -c "from lemas_tts_amd import build; build.build_library(force=True)" ;;
    mbuild)     # measurement build WITHOUT the phase stamps (engine options of the kept experiments, e.g. attn_f8qk); `pbuild` restores the product
      LEMAS_EXTRA_HIPCC_FLAGS=-DLEMAS_MEASUREMENT_BUILD timeout 900 python -c "from lemas_tts_amd import build; build.build_library(force=True)" ;;
    pbuild)
      timeout 900 python -c "from lemas_tts_amd import build; build.build_library(force=True)" ;;
    kbench)
      timeout 900 python tools/kbench.py $(sp "${rest:-gemm}") 2>&1 | grep -v amdgpu.ids | tee -a "$O/${TAG}_kbench.txt" ;;
    scale)
      n=${rest:-8}
      LEMAS_SHARE_GPU=1 LEMAS_DIST_BACKEND=gloo timeout 900 python bench.py --gpus $n --no-cpu-baseline --steps 2 --warmup 1 > "$O/${TAG}_bench_${n}ranks_shared_gpu.json" 2>> "$O/${TAG}_bench.err"
      cut -c1-260 "$O/${TAG}_bench_${n}ranks_shared_gpu.json"
      LEMAS_SHARE_GPU=1 LEMAS_DIST_BACKEND=gloo timeout 900 python bench.py --gpus $n --job configs3_full --steps 1 --warmup 1 > "$O/${TAG}_job_${n}ranks_shared_gpu.json" 2>> "$O/${TAG}_bench.err"
      cut -c1-260 "$O/${TAG}_job_${n}ranks_shared_gpu.json" ;;
    pyprof)
      name=${rest%%:*}; cmd=$(sp "${rest#*:}"); rm -rf /tmp/pyprof_$name
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pyprof_$name -- python $R/$cmd > /tmp/pyprof_$name.out 2> /tmp/pyprof_$name.log)
      python tools/rocpd_summary.py "$(find /tmp/pyprof_$name -name '*_results.db' | head -1)" > "$O/${TAG}_kernel_stats_$name.txt"
      [ -n "${SEQ:-}" ] && python tools/rocpd_sequence.py "$(find /tmp/pyprof_$name -name '*_results.db' | head -1)" $SEQ > "$O/${TAG}_sequence_$name.txt"
      grep -v amdgpu.ids /tmp/pyprof_$name.out | tail -1 > "$O/${TAG}_under_rocprof_$name.json"; head -24 "$O/${TAG}_kernel_stats_$name.txt" | cut -c1-150 ;;
    pypmc)
      name=${rest%%:*}; cmd=$(sp "${rest#*:}"); : > "$O/${TAG}_pmc_$name.txt"
      for grp in "${PMC_GROUPS[@]}" "FETCH_SIZE" "WRITE_SIZE"; do
        d=/tmp/pypmc_${name}_$(echo $grp | tr ' ' '_' | cut -c1-40); rm -rf $d
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $d -- python $R/$cmd > /dev/null 2> /tmp/pmc_err.log) || tail -3 /tmp/pmc_err.log
        echo "## $grp" >> "$O/${TAG}_pmc_$name.txt"
        python tools/rocpd_pmc.py "$(find $d -name '*_results.db' | head -1)" mdx_ gemm_f32 >> "$O/${TAG}_pmc_$name.txt"
      done; tail -12 "$O/${TAG}_pmc_$name.txt" ;;
    py)
      timeout 1500 python $(sp "$rest") 2>&1 | grep -v amdgpu.ids | tee -a "$O/${TAG}_py.txt" ;;
    *) echo "unknown step $step" ;;
  esac
done
tail -5 "$O/${TAG}_bench.err" 2>/dev/null
