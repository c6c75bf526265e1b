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
    assert one["roofline_vocoder"]["bound"] == "mfma_f32" and 0 < one["roofline_vocoder"]["hbm_view"]["frac"] < 1 and 0 < one["roofline_vocoder"]["frac"] < 1 and one["hoist_ms"] > 0
    assert set(one["phase_ms"]) >= {"hoists", "step_loop", "vocoder", "d2h"} and "vocoder" in one["kernel_time_share_utterance"]
    two = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--depth", "2", "--no-cpu-baseline", "--workload", "short"],
                 env={"LEMAS_SHARE_GPU": "1", "LEMAS_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["parallelism"].startswith("dp2") and two["scaling"] == "weak"
    assert two["config"]["audio_seconds_per_step"] == one["config"]["audio_seconds_per_step"]      # fixed work per GPU


@pytest.mark.timeout(900)
def test_bench_under_the_drivers_own_launcher():
    """N > 1 the way the driver starts it: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`` (two ranks sharing the box's one GPU, gloo for the collectives): exactly ONE JSON line on
    the launcher's stdout, from rank 0, whole-job value."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--depth", "2",
           "--no-cpu-baseline", "--workload", "short"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, LEMAS_SHARE_GPU="1", LEMAS_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["steps"] == 1 and two["warmup"] == 1 and two["scaling"] == "weak" and two["value"] > 0
    assert len(two["per_rank_ms"]["all"]) == 2 and two["config"]["utterances_total"] == 2


@pytest.mark.timeout(900)
def test_bench_runs_every_rccl_call_of_the_multi_gpu_path_in_a_world_of_one():
    """LEMAS_FORCE_DIST=1 sends ``--gpus 1`` through the N > 1 code: RCCL process group bound to the device, the flat weight
    broadcast INTO DEVICE MEMORY, device-to-device engine loads from views of that buffer, the barriers / all_gather / all_reduce
    around the timed region -- and the utterance it times is checked against the reference's own output (full depth, full NFE)."""
    line = _bench(["--steps", "1", "--warmup", "1", "--no-cpu-baseline"], env={"LEMAS_FORCE_DIST": "1", "LEMAS_DIST_BACKEND": "nccl"})
    wb = line["weight_broadcast"]
    # the DiT blocks' GEMM weights travel in bf16 on the bf16 path (1.03 GB instead of 1.40 with the vocoder; bit-identical results: the mel-MSE below is the
    # single-process run's to the last digit)
    assert wb["backend"] == "nccl" and wb["on_device"] is True and wb["world"] == 1 and wb["block_gemm_weights"] == "bf16" and 0.98e9 < wb["bytes"] < 1.08e9
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


@pytest.mark.timeout(900)
def test_bench_sharded_job_two_ranks_on_one_gpu():
    """``--job configs3_full``: rank 0 owns the 64 utterances, deals them (shard_utterances), the ranks sample + vocode their shards in
    batches of 8 and the waveforms come back to rank 0's host (precedent uvr5/multiprocess_cuda_infer.py:404-420).  Two
    ranks share the one GPU over gloo; reduced depth.  bench.py itself asserts that an utterance alone equals the utterance in its batch."""
    two = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--depth", "2", "--job", "configs3_full"],
                 env={"LEMAS_SHARE_GPU": "1", "LEMAS_DIST_BACKEND": "gloo"})
    c = two["config"]
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and c["workload_key"] == "configs3_full"
    assert c["utterances_total"] == 64 and c["utterances_per_rank"] == 32 and c["utterances_per_batch"] == 8 and c["batches_per_rank"] == 4
    assert c["utterance_alone_equals_in_batch"] is True and c["audio_seconds_total"] == pytest.approx(64 * 8.0, rel=1e-6)
    pr = two["per_rank_ms"]
    assert len(pr["compute"]) == 2 and min(pr["compute"]) > 0 and two["job_ms"] >= max(pr["compute"]) and two["value"] > 0
    one = _bench(["--steps", "1", "--warmup", "1", "--depth", "2", "--job", "configs3_full"])
    assert one["n_gpus"] == 1 and one["config"]["batches_per_rank"] == 8 and one["per_rank_ms"]["scatter"][0] >= 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("wl", ["configs2", "configs4"])
def test_bench_times_the_ragged_prosody_batch_and_the_speech_edit_case(wl):
    """the two BASELINE configurations beside the headline that bench.py times since round 4 (reduced depth here: launch shapes, vocoder
    segments and bookkeeping; the full-depth parity of the same fixtures is tests/test_gpu_06_configs.py and the full bench run)"""
    line = _bench(["--steps", "1", "--warmup", "1", "--depth", "2", "--no-cpu-baseline", "--no-clock-power", "--workload", wl])
    c = line["config"]
    assert c["workload_key"] == wl and line["value"] > 0 and line["roofline"]["frac"] > 0
    if wl == "configs2":
        assert c["utterances_per_gpu_per_step"] == 8 and c["waveforms_per_step"] == 8 and c["vocode"] == "each"
        assert c["real_frames_per_step"] == 10910 and c["rows_computed_per_step"] == 15360 and c["padded_row_waste"] == pytest.approx(1 - 10910 / 15360)
        assert line["dtype"] == "bf16" and "NFE=32" in line["metric"]
    else:
        assert c["waveforms_per_step"] == 1 and c["vocode"] == "whole" and c["audio_seconds_per_step"] == pytest.approx(256 * 2813 / 24000)
        assert line["dtype"] == "fp8" and "NFE=48" in line["metric"]


@pytest.mark.timeout(1500)
def test_eight_ranks_share_the_one_gpu_weak_line_and_sharded_job():
    """The 8-GPU run the driver launches, rehearsed end to end with EIGHT ranks on the one GPU over gloo (reduced depth): eight processes
    each broadcasting / loading weights, capturing graphs and replaying 32 steps per utterance from one host at once, then the sharded job
    with 7 peers' scatter and gather in flight together (parallel.scatter_from_rank0 / gather_to_rank0 are non-blocking).  Rank 0 also
    reports the roofline of its kernels for N > 1."""
    env = {"LEMAS_SHARE_GPU": "1", "LEMAS_DIST_BACKEND": "gloo"}
    eight = _bench(["--gpus", "8", "--steps", "1", "--warmup", "1", "--depth", "2", "--no-cpu-baseline", "--workload", "configs3"], env=env)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "weak" and eight["value"] > 0
    assert eight["config"]["utterances_total"] == 64 and len(eight["per_rank_ms"]["all"]) == 8
    assert eight["weight_broadcast"]["world"] == 8 and eight["weight_broadcast"]["block_gemm_weights"] == "bf16"
    assert "roofline" in eight and eight["roofline"]["frac"] > 0 and eight["roofline"]["bound"] in ("mfma", "hbm")
    job = _bench(["--gpus", "8", "--steps", "1", "--warmup", "1", "--depth", "2", "--job", "configs3_full"], env=env)
    c = job["config"]
    assert job["n_gpus"] == 8 and c["utterances_total"] == 64 and c["utterances_per_rank"] == 8 and c["batches_per_rank"] == 1
    assert c["utterance_alone_equals_in_batch"] is True and len(job["per_rank_ms"]["compute"]) == 8
