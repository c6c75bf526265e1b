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
    two = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--depth", "2", "--no-cpu-baseline", "--workload", "short"],
                 env={"LEMAS_SHARE_GPU": "1", "LEMAS_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["parallelism"].startswith("dp2") and two["scaling"] == "weak"
    assert two["config"]["audio_seconds_per_step"] == one["config"]["audio_seconds_per_step"]      # fixed work per GPU


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
