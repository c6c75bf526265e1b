"""GPU tier: bench.py's launch paths -- the single-GPU line and the N > 1 path started by bench.py itself (rehearsed with two ranks
sharing the one GPU over gloo), at reduced depth so it takes seconds.  The driver runs the real thing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_spawns_its_own_ranks():
    one = _bench(["--steps", "1", "--warmup", "1", "--depth", "2", "--no-cpu-baseline", "--workload", "short"])
    assert one["n_gpus"] == 1 and one["value"] > 0 and "roofline" in one and one["roofline"]["frac"] > 0
    assert one["config"]["workload_key"] == "short" and one["config"]["parallelism"].startswith("dp1")
    assert one["roofline_vocoder"]["bound"] == "hbm" and 0 < one["roofline_vocoder"]["frac"] < 1 and one["hoist_ms"] > 0
    assert set(one["phase_ms"]) >= {"hoists", "step_loop", "vocoder", "d2h"} and "vocoder" in one["kernel_time_share_utterance"]
    two = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--depth", "2", "--no-cpu-baseline", "--workload", "short"],
                 env={"LEMAS_SHARE_GPU": "1", "LEMAS_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["parallelism"].startswith("dp2") and two["scaling"] == "weak"
    assert two["config"]["audio_seconds_per_step"] == one["config"]["audio_seconds_per_step"]      # fixed work per GPU


@pytest.mark.timeout(900)
def test_bench_runs_every_rccl_call_of_the_multi_gpu_path_in_a_world_of_one():
    """LEMAS_FORCE_DIST=1 sends ``--gpus 1`` through the N > 1 code: RCCL process group bound to the device, the flat weight
    broadcast INTO DEVICE MEMORY, device-to-device engine loads from views of that buffer, the barriers / all_gather / all_reduce
    around the timed region -- and the utterance it times is checked against the reference's own output (full depth, full NFE)."""
    line = _bench(["--steps", "1", "--warmup", "1", "--no-cpu-baseline"], env={"LEMAS_FORCE_DIST": "1", "LEMAS_DIST_BACKEND": "nccl"})
    wb = line["weight_broadcast"]
    assert wb["backend"] == "nccl" and wb["on_device"] is True and wb["world"] == 1 and wb["bytes"] > 1.3e9
    assert line["n_gpus"] == 1 and line["mel_mse_vs_reference"] is not None and line["mel_mse_vs_reference"] <= 1e-4
    assert line["per_rank_ms"]["min"] > 0 and len(line["per_rank_ms"]["all"]) == 1
    assert line["config"]["utterances_total"] == 1 and line["config"]["utterances_timed"] == 1
    assert line["config"]["parity_fixture"] == "configs1_nfe32.npz"


@pytest.mark.timeout(900)
def test_bench_configs3_reports_the_utterance_count_of_the_whole_job():
    two = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--depth", "2", "--no-cpu-baseline", "--workload", "configs3"],
                 env={"LEMAS_SHARE_GPU": "1", "LEMAS_DIST_BACKEND": "gloo"})
    assert two["config"]["utterances_total"] == 16 and two["config"]["utterances_timed"] == 16          # 8 per GPU: 64 on 8 GPUs
    assert two["config"]["per_rank_audio_seconds"] == [pytest.approx(8 * 8.0, rel=1e-6)] * 2
    assert len(two["per_rank_ms"]["all"]) == 2 and two["per_rank_ms"]["max"] >= two["per_rank_ms"]["min"] > 0
    assert two["config"]["audio_seconds_per_rank"] == pytest.approx(8 * 8.0, rel=1e-6)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
