# In-situ timeline of one ODE step (tools/timeline_step.py): measurement build -> run -> product build.  On the GPU box only.
cd $GRAFT_REPO_ROOT
LEMAS_EXTRA_HIPCC_FLAGS=-DLEMAS_PHASE_TIMESTAMPS python -c "from lemas_tts_amd import build; build.build_library(force=True)"
python tools/timeline_step.py --workload configs1 2>&1 | grep -v amdgpu.ids
python tools/timeline_step.py --workload configs3 2>&1 | grep -v amdgpu.ids
python -c "from lemas_tts_amd import build; build.build_library(force=True)"
