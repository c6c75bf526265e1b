#!/usr/bin/env python
"""Headline benchmark of the MI355X-native LEMAS-TTS acoustic path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload configs1|configs3|short]

With ``--gpus N`` (N > 1) and no launcher environment, bench.py starts the N ranks ITSELF (one process per GPU, the reference's
own multi-GPU precedent: uvr5/multiprocess_cuda_infer.py:404-420); started under ``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`` it is one of the N ranks.  Either way the
world size must equal ``--gpus``.

Metric (BASELINE.json): audio-seconds/sec @24 kHz, NFE=32, CFG on.  A "step" is ONE pass of the hot path over one batch of
synthetic input = one utterance batch per GPU: per-utterance hoists (text embedding, conditioning projection, time/AdaLN
tables -- recomputed every utterance, nothing is cached across steps) + 32 Euler steps of the CFG-folded DiT + Vocos decode
of the generated frames + device->host copy of the waveform.  Inputs (reference mel, token ids, noise) are resident in HBM
when the timed region starts; model load is outside it.

Workloads (per GPU and step; fixed as N grows = weak scaling):
  configs1  BASELINE configs[1] (the headline): multilingual_grl, batch 1, 10 s reference + 10 s target (F=938, N=1875)
  configs3  BASELINE configs[3]'s per-GPU share: 8 utterances of 4 s reference + 8 s target (F=375, N=1125) as one batch
            (64 utterances over 8 GPUs)
  short     one short utterance, 4 s + 4 s (F=375, N=750): what the entry script mostly sees
All: NFE 32, cfg 2.0, sway coef 5 (capped 3.486), bf16 MFMA operands with fp32 accumulate / residual / ODE state.
Multi-GPU: utterance-level data parallelism, weights generated on rank 0 and broadcast over RCCL/xGMI into device memory
(loaded device-to-device on every rank), no collective in the step loop.

CORRECTNESS inside the run: the mel the timed region produced last is compared with the committed output of the REFERENCE
itself on the same inputs (tests/golden/configs1_nfe32.npz, configs3_share_nfe32.npz: 22 blocks, all 32 steps; the short
workload against configs0_nfe16.npz, the same utterance over 16 steps, in one extra untimed solve; made by oracle/gen_golden.py
--full-size); the run FAILS above mel-MSE 1e-4 and the value is reported as ``mel_mse_vs_reference``.

Pipelining (``--overlap 1``, the default): the Vocos decode + D2H of utterance i run on a side stream under the step loop of
utterance i+1 (the step loop itself still holds ONE utterance batch at a time: the B = 1 definition of configs[1] is unchanged;
every waveform is on the host when the timed region ends).  ``--overlap 0`` serialises them as the reference's loop does.

The JSON line also carries
  roofline     -- the dominant kernel BY SYMBOL (what rocprofv3 --stats lists; out-proj and FF2 share one instantiation):
                  algorithmic FLOPs per launch / average launch duration, measured live with HIP event pairs stamped by the
                  dispatch itself (hipExtLaunchKernelGGL) in a short eager pass with the timed region's launch shapes, vs
                  2.5 PFLOP/s dense bf16; ``roofline_gemm_family`` = all block GEMMs together, ``path_frac`` = whole path;
  roofline_vocoder -- the HBM-bound phase: algorithmic bytes of one decode (fp32 weights + 400 L + 1024 (L-1)) over
                  its measured duration, vs 8 TB/s; ``phase_ms`` = hoists / step loop / vocoder / D2H of one utterance, measured
                  serially with stream events after the timed region;
  cpu_baseline -- the fp32 oracle (oracle/lemas_oracle.py, a port of the reference's path) timed on this box's host
                  cores on a bounded sample (1 of the 32 Euler steps at full N, scaled x32, + the full vocoder).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lemas_tts_amd import synth  # noqa: E402
from lemas_tts_amd.model.layout import DiTArch  # noqa: E402

VOCAB = 898
NFE, CFG, SWAY = 32, 2.0, 5
HOP, SR = 256, 24000
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_FP8_PEAK_TFLOPS = 5000.0
MEL_MSE_TOL = 1e-4
GOLDEN = os.path.join(ROOT, "tests", "golden")

WORKLOADS = {
    "configs1": dict(B=1, F=938, N=1875, golden="configs1_nfe32.npz", golden_steps=32,
                     desc="BASELINE configs[1]: multilingual_grl, batch 1, 10 s ref + 10 s target"),
    "configs3": dict(B=8, F=375, N=1125, golden="configs3_share_nfe32.npz", golden_steps=32,
                     desc="BASELINE configs[3] per-GPU share: multilingual_grl, 8 utterances of 4 s ref + 8 s target as one batch"),
    "short": dict(B=1, F=375, N=750, golden="configs0_nfe16.npz", golden_steps=16,
                  desc="one short utterance: multilingual_grl, batch 1, 4 s ref + 4 s target"),
}


def fwd_flops(B: int, N: int) -> float:
    """FLOPs of one DiT forward with the AdaLN/time GEMVs hoisted."""
    return B * (378_888_192.0 * N + 90_112.0 * N * N)


def class_flops(cls: str, rows: int, n: int, bb: int, d: int = 1024, ff: int = 2048, heads: int = 16) -> float:
    """Algorithmic FLOPs of one launch of a step-loop kernel class (rows = real frames of the launch, not the padded row space)."""
    return {"gemm_qkv_fused": 2.0 * rows * 3 * d * d, "gemm_qk_rope": 2.0 * rows * 2 * d * d, "gemm_v_t": 2.0 * rows * d * d,
            "gemm_attn_out": 2.0 * rows * d * d, "gemm_ff1_gelu": 2.0 * rows * ff * d, "gemm_ff2": 2.0 * rows * d * ff,
            "attention": 4.0 * n * n * 64 * bb * heads}[cls]


# kernel classes of the profile pass -> the symbol rocprofv3 lists them under (out-proj and FF2 are ONE instantiation)
SYMBOL = {"gemm_qkv_fused": "gemm_qkv_fused_kernel", "gemm_qk_rope": "gemm_bf16_kernel<EPI_QK_ROPE>", "gemm_v_t": "gemm_bf16_kernel<EPI_V_T>",
          "gemm_attn_out": "gemm_bf16_kernel<EPI_GATE_RES>", "gemm_ff2": "gemm_bf16_kernel<EPI_GATE_RES>",
          "gemm_ff1_gelu": "gemm_bf16_kernel<EPI_BIAS_GELU>", "attention": "attn_fwd_splitkv_kernel"}


def build_inputs(w: dict, rank: int, device):
    """rank 0 runs the utterance(s) of the committed reference fixture (so the result can be checked); the others seeded ones"""
    B, F, N = w["B"], w["F"], w["N"]
    fx = None
    if w["golden"] and os.path.exists(os.path.join(GOLDEN, w["golden"])):
        fx = synth.expand_reference_fixture(dict(np.load(os.path.join(GOLDEN, w["golden"]))))
    if fx is not None and rank == 0:
        cond, text, y0 = torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]), torch.from_numpy(fx["y0"])
        assert tuple(cond.shape) == (B, F, 100) and tuple(y0.shape) == (B, N, 100)
    else:
        nt = round(N * 0.17)
        cond = torch.stack([torch.from_numpy(synth.synth_cond_mel(1234 + 97 * rank + b, F)) for b in range(B)])
        text = torch.stack([torch.from_numpy(synth.synth_tokens(1234 + 97 * rank + b, nt, VOCAB)) for b in range(B)])
        y0 = torch.stack([torch.from_numpy(synth.synth_noise(1234 + 97 * rank + b, N)) for b in range(B)])
    return cond.to(device), text.to(device), y0.to(device), (fx if rank == 0 else None)


def mel_mse(out: torch.Tensor, ref: np.ndarray, F: int) -> float:
    d = (out.detach().cpu().double() - torch.from_numpy(ref).double())[:, F:]
    return float((d ** 2).mean())


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box exposes 256
    logical CPUs but the container is throttled to a quota; asking torch for 256 threads there is 50x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(sd, vsd, arch, w):
    """Oracle on host cores, bounded sample of the same workload."""
    from oracle import lemas_oracle as O  # checker / baseline only
    cores = usable_cores()
    torch.set_num_threads(cores)
    F, N = w["F"], w["N"]
    cond = torch.from_numpy(synth.synth_cond_mel(1234, F))[None]
    text = torch.from_numpy(synth.synth_tokens(1234, round(N * 0.17), VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(1234, N))[None]
    cfm = O.OracleCFM(sd, arch)
    sub = 1
    tg = O.time_grid(NFE, SWAY)[: sub + 1]
    t0 = time.perf_counter()
    out, _ = cfm.sample(cond, text, N, y0=y0, steps=sub, cfg_strength=CFG, sway_sampling_coef=SWAY, t_grid=tg)
    t_steps = time.perf_counter() - t0
    mel = out[:, F - 1:, :].permute(0, 2, 1)
    t0 = time.perf_counter()
    O.OracleVocos(vsd).decode(mel)
    t_voc = time.perf_counter() - t0
    audio_s = HOP * (mel.shape[-1] - 1) / SR
    est = t_steps * (NFE / sub) + t_voc
    return {"value": audio_s / est, "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "sample": f"one utterance of the workload (F={F}, N={N}): {sub} of {NFE} Euler steps ({t_steps:.1f} s) scaled x{NFE // sub} "
                      f"+ full Vocos decode ({t_voc:.2f} s); fp32 torch, {cores} threads"}


def clock_power(step_fn, seconds: float = 4.0):
    """Engine clock and package power WHILE the timed loop's work runs (an untimed repeat of it), read from ``rocm-smi`` in a second
    process: the board manages the clock against a 1400 W package limit, and the solve runs ~9 % under the 2.4 GHz the MFMA peak is
    quoted at (profiles/r03_power_clocks.txt).  None when rocm-smi is missing or prints something else."""
    import re
    import shutil
    import subprocess
    import threading
    import time
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    samples, stop = [], threading.Event()

    def sampler():
        time.sleep(1.0)                                  # let the clock settle
        while not stop.is_set():
            try:
                out = subprocess.run([exe, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            except Exception:
                return
            m1 = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
            m2 = re.search(r"Package Power \(W\): ([0-9.]+)", out)
            if m1 and m2:
                samples.append((int(m1.group(1)), float(m2.group(1))))
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.time()
    while time.time() - t0 < seconds:
        step_fn()
        torch.cuda.synchronize()
    stop.set()
    th.join(timeout=30)
    busy = [x for x in samples if x[0] > 500]             # a sample taken after the loop ended reads the idle clock
    if not busy:
        return None
    return {"sclk_mhz": sum(x[0] for x in busy) / len(busy), "package_w": sum(x[1] for x in busy) / len(busy), "samples": len(busy),
            "sclk_ceiling_mhz": 2400, "package_limit_w": 1400,
            "source": "rocm-smi --showclocks --showpower from a thread while the timed loop's work repeats (untimed)"}


def spawn_ranks(n: int) -> int:
    """Start the n ranks of a one-node data-parallel run ourselves (torchrun's environment contract, rendezvous on 127.0.0.1)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="configs1")
    ap.add_argument("--depth", type=int, default=22, help="DiT depth (22 = the shipped model; smaller only for debugging, no parity check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", type=int, default=0, help="1 = block GEMMs on the fp8-e4m3 (MXFP8) path of BASELINE config 5; "
                    "NOT the headline configuration (configs[1] is bf16): the line is then labelled dtype fp8")
    ap.add_argument("--dual", type=int, default=1, help="1 = CFG branches as two concurrent lanes (default), 0 = one stream")
    ap.add_argument("--overlap", type=int, default=1, help="1 = Vocos decode + D2H of utterance i on a side stream under the step loop of "
                    "utterance i+1 (default), 0 = strictly serial")
    ap.add_argument("--ln-fused", type=int, default=-1, help="engine option ln_fused (-1 = engine default)")
    ap.add_argument("--ln-fold", type=int, default=-1, help="engine option ln_fold (-1 = engine default)")
    ap.add_argument("--no-clock-power", action="store_true", help="skip the rocm-smi clock / power sampling pass (profiler runs)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))

    # stdout carries ONE line, the JSON record: everything else that lands on file descriptor 1 from here on (the RCCL / Gloo banners
    # are printed by C code) goes to stderr, and the record is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # LEMAS_FORCE_DIST=1: run the N > 1 code path (process group, weight broadcast into device memory, device-to-device engine
    # load, collectives around the timed region) with a world of ONE -- every RCCL call of the 8-GPU run, executable on a 1-GPU box
    force_dist = os.environ.get("LEMAS_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU path)"
    # LEMAS_DIST_BACKEND=gloo + LEMAS_SHARE_GPU=1 exist only to rehearse the N>1 code path on a 1-GPU box
    backend = os.environ.get("LEMAS_DIST_BACKEND", "nccl")
    if os.environ.get("LEMAS_SHARE_GPU") == "1":
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} are visible")
    dist = None
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    from lemas_tts_amd.parallel import pin_to_gpu_numa_node  # noqa: E402
    affinity = pin_to_gpu_numa_node(local_rank) if use_dist else None     # one rank per GPU: host threads next to that GPU's NUMA node
    if use_dist:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_PORT" not in os.environ:       # world of one without a launcher
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == a.gpus
    comm_device = device if (not use_dist or backend == "nccl") else torch.device("cpu")

    from lemas_tts_amd.engine import VocosEngine  # noqa: E402
    from lemas_tts_amd.model.cfm import CFM, time_grid  # noqa: E402
    from lemas_tts_amd.parallel import broadcast_state_dict  # noqa: E402

    w = WORKLOADS[a.workload]
    B, F_REF, N_TOT = w["B"], w["F"], w["N"]
    arch = DiTArch(depth=a.depth)
    # weights: generated on rank 0, broadcast over RCCL/xGMI; every rank loads them device-to-device from the received buffer
    sd = synth.synth_cfm_state_dict(arch, VOCAB, 1234) if rank == 0 else None
    vsd = synth.synth_vocos_state_dict(1234) if rank == 0 else None
    sd_host, vsd_host = sd, vsd
    bcast = None
    if use_dist:
        t_b = time.perf_counter()
        sd = broadcast_state_dict(sd, arch, VOCAB, comm_device, dist)
        vsd = broadcast_state_dict(vsd, None, None, comm_device, dist, vocos=True)
        nbytes = 4 * (sum(int(np.prod(v.shape)) for v in sd.values()) + sum(int(np.prod(v.shape)) for v in vsd.values()))
        bcast = {"backend": backend, "bytes": nbytes, "seconds": time.perf_counter() - t_b,
                 "on_device": bool(comm_device.type == "cuda"), "world": world}
    model = CFM(arch, VOCAB, sd, device=device)
    model.engine.set_option("dual", a.dual)
    model.engine.set_option("fp8", a.fp8)
    if a.ln_fused >= 0:
        model.engine.set_option("ln_fused", a.ln_fused)
    if a.ln_fold >= 0:
        model.engine.set_option("ln_fold", a.ln_fold)
    model.engine.set_option("table_cache", 0)      # hoists are redone for every utterance: nothing cached across steps
    vocoder = VocosEngine(vsd, device=device)
    del sd, vsd                                     # the engines own their copies; the broadcast buffer can go
    cond, text, y0, fx = build_inputs(w, rank, device)
    text = text.cpu().pin_memory()                 # token ids arrive from the host frontend (api.py:201-204): no D2H sync per utterance
    L_GEN = N_TOT - F_REF + 1
    host_wav = torch.empty((B, HOP * (L_GEN - 1)), dtype=torch.float32).pin_memory()
    last = {}
    side = torch.cuda.Stream(device) if a.overlap else None

    def vocode(out):
        # the driver vocodes generated[:, nw // 256:] = frames F-1.. (utils_infer.py:520,546): L_gen = N - F + 1
        wav = vocoder.decode(out[:, F_REF - 1:, :].permute(0, 2, 1))
        host_wav.copy_(wav, non_blocking=True)
        return wav

    def step(steps=NFE):
        out, _ = model.sample(cond, text, N_TOT, steps=steps, cfg_strength=CFG, sway_sampling_coef=SWAY, y0=y0, use_acc_grl=False)
        last["out"] = out
        if side is None:
            return vocode(out)
        # decode + D2H of THIS utterance on the side stream; the next utterance's hoists and step loop follow on the main one
        done = torch.cuda.Event()
        done.record()
        with torch.cuda.stream(side):
            side.wait_event(done)
            out.record_stream(side)
            return vocode(out)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [1e3 * elapsed / max(a.steps, 1)]
    if dist:
        mine = torch.tensor([elapsed], device=comm_device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                       # load imbalance stays visible: per-rank step times, not only the max
        per_rank_ms = [1e3 * float(t.item()) / max(a.steps, 1) for t in every]
        tmax = mine.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert np.isfinite(host_wav.numpy()).all()

    # ---- correctness of what was just timed (rank 0 holds the fixture's utterance): reference output on the same inputs
    mse = None
    if rank == 0 and fx is not None and a.depth == 22 and a.steps + a.warmup > 0:
        if w["golden_steps"] == NFE:
            mse = mel_mse(last["out"], fx["out"], F_REF)             # the timed region's own last result
        else:                                                        # the fixture is a short solve: one extra untimed sample
            out, _ = model.sample(cond, text, N_TOT, steps=w["golden_steps"], cfg_strength=CFG, sway_sampling_coef=SWAY, y0=y0, use_acc_grl=False)
            mse = mel_mse(out, fx["out"], F_REF)
        tol = MEL_MSE_TOL
        assert mse <= tol, f"bench.py: the timed path is WRONG: mel-MSE vs the reference's output {mse:.3e} > {tol:g}"

    result = None
    if rank == 0:
        audio_per_step = B * HOP * (L_GEN - 1) / SR
        value = world * a.steps * audio_per_step / elapsed
        flops_step = 2 * NFE * fwd_flops(B, N_TOT) * (a.depth / 22.0)
        path_tflops = flops_step / (elapsed / a.steps) / 1e12
        result = {
            "metric": "audio-seconds/sec @24kHz (NFE=32, CFG on)", "value": value, "unit": "audio-seconds/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8" if a.fp8 else "bf16", "data": "synthetic",
            "rtf": elapsed / (a.steps * audio_per_step),
            "mel_mse_vs_reference": mse,
            "config": {"workload": f"{w['desc']} (F={F_REF}, N={N_TOT}), NFE={NFE}, CFG={CFG}, sway coef {SWAY} (capped), "
                                   + ("fp8-e4m3 (MXFP8) GEMM operands, bf16 attention" if a.fp8 else "bf16 MFMA operands")
                                   + " / fp32 state, Vocos decode + D2H included",
                       "workload_key": a.workload,
                       "utterances_per_gpu_per_step": B, "audio_seconds_per_step": audio_per_step,
                       "utterances_total": world * B, "utterances_timed": world * B * a.steps,     # the job's utterances per step / in the timed region
                       "audio_seconds_per_rank": a.steps * audio_per_step,
                       "per_rank_audio_seconds": [a.steps * audio_per_step] * world,                  # weak scaling: the same work on every rank
                       "pipelining": ("vocoder + D2H of utterance i on a side stream under the step loop of utterance i+1"
                                      if a.overlap else "none (strictly serial)"),
                       "parallelism": f"dp{world} ({world} process(es), one per GPU; utterance sharding, RCCL weight broadcast into "
                                      "device memory, no step-loop collectives)",
                       "depth": a.depth, "weights": "synthetic N(0,0.02^2), seed 1234",
                       "parity_fixture": w["golden"] if mse is not None else None},
            "path_tflops": path_tflops,
            "path_frac": path_tflops / (MFMA_BF16_PEAK_TFLOPS if not a.fp8 else MFMA_FP8_PEAK_TFLOPS),
            "per_rank_ms": {"min": min(per_rank_ms), "max": max(per_rank_ms), "all": [round(x, 3) for x in per_rank_ms]},
        }
        if bcast is not None:
            result["weight_broadcast"] = bcast
        if affinity is not None:
            result["config"]["cpu_affinity"] = affinity

    # ---- roofline of the dominant kernel: short eager pass with per-launch HIP events (rank 0, N == 1 only)
    if rank == 0 and world == 1:
        eng = model.engine
        eng.set_option("profile", 1)
        sub = 4
        tg = time_grid(NFE, SWAY)[: sub + 1]
        cm = torch.zeros(B, N_TOT, dtype=torch.bool)
        cm[:, :F_REF] = True
        eng.prepare(torch.nn.functional.pad(cond, (0, 0, 0, N_TOT - F_REF)), cm, text, tg.numpy(), cond_frames=F_REF, cfg_strength=CFG)
        eng.solve(y0, want_out=False)
        prof = eng.profile_read()
        eng.set_option("profile", 0)
        lanes = 2 if a.dual else 1                 # dual: each launch covers one CFG branch (B of the 2B branch-rows)
        rows, bb = 2 * B * N_TOT // lanes, 2 * B // lanes
        mm = {k: v for k, v in prof.items() if k in SYMBOL and v[1] > 0}
        by_sym = {}
        for k, (ms, cnt) in mm.items():
            e = by_sym.setdefault(SYMBOL[k], {"ms": 0.0, "launches": 0, "flops": 0.0, "classes": []})
            e["ms"] += ms; e["launches"] += int(cnt); e["flops"] += class_flops(k, rows, N_TOT, bb) * cnt; e["classes"].append(k)
        dom = max(by_sym, key=lambda s: by_sym[s]["ms"])      # dominant kernel = the symbol with the largest total time
        d = by_sym[dom]
        avg_us = 1e3 * d["ms"] / d["launches"]
        fl = d["flops"] / d["launches"]
        ach = fl / (avg_us * 1e-6) / 1e12
        total_ms = sum(v[0] for v in prof.values())
        # HBM-side bytes per launch and the rocprofv3 average of the same symbol come from COMMITTED profiler passes over this very
        # command (rocprofv3 cannot run inside bench.py); both are labelled with the file they were read from
        traffic, traffic_src, rocprof_us, rocprof_src = None, None, None, None
        for name in ("r03_traffic.json", "r02_traffic.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", name)))
                traffic, traffic_src = tj[a.workload][dom]["hbm_bytes_per_launch"], f"profiles/{name} <- {tj.get('_source', '?')}"
                break
            except (OSError, KeyError, ValueError, TypeError):
                continue
        try:
            kj = json.load(open(os.path.join(ROOT, "profiles", "r03_kernel_avgs.json")))
            rocprof_us, rocprof_src = kj[a.workload][dom]["avg_us"], f"profiles/r03_kernel_avgs.json <- {kj.get('_source', '?')}"
        except (OSError, KeyError, ValueError, TypeError):
            pass
        is_gemm = dom != "attn_fwd_splitkv_kernel"
        peak = MFMA_FP8_PEAK_TFLOPS if (a.fp8 and is_gemm) else MFMA_BF16_PEAK_TFLOPS   # attention stays bf16
        result["roofline"] = {"bound": "mfma", "kernel": dom, "classes": d["classes"], "achieved": ach, "peak": peak,
                              "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                              "avg_launch_us": avg_us, "avg_launch_us_source": "this run: eager pass, per-launch HIP event pairs, lanes serialised",
                              "avg_launch_us_rocprof": rocprof_us, "avg_launch_us_rocprof_source": rocprof_src,
                              "frac_rocprof": (fl / (rocprof_us * 1e-6) / 1e12 / peak) if rocprof_us else None,
                              "launches": d["launches"], "flops_per_launch": fl,
                              "time_share": d["ms"] / total_ms,
                              "launch_shape": f"{bb} x {N_TOT} frames per launch ({lanes} concurrent lane(s) in the timed region)"}
        g_ms = sum(v["ms"] for s, v in by_sym.items() if s != "attn_fwd_splitkv_kernel")
        g_fl = sum(v["flops"] for s, v in by_sym.items() if s != "attn_fwd_splitkv_kernel")
        gpeak = MFMA_FP8_PEAK_TFLOPS if a.fp8 else MFMA_BF16_PEAK_TFLOPS
        result["roofline_gemm_family"] = {"achieved": g_fl / (g_ms * 1e-3) / 1e12, "peak": gpeak, "frac": g_fl / (g_ms * 1e-3) / 1e12 / gpeak,
                                          "time_share": g_ms / total_ms, "unit": "TFLOP/s"}
        result["kernel_tflops"] = {k: round(class_flops(k, rows, N_TOT, bb) / (1e3 * v[0] / v[1] * 1e-6) / 1e12, 1) for k, v in mm.items()}
        result["kernel_time_share"] = {k: round(v[0] / total_ms, 4) for k, v in prof.items() if v[1] > 0}
        result["kernel_avg_us"] = {k: round(1e3 * v[0] / v[1], 2) for k, v in prof.items() if v[1] > 0}

        # ---- the other phases of one utterance, serially, with stream events: hoists / step loop / vocoder / D2H
        def timed_ms(fn, reps=3):
            best = None
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                r = fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                best = ms if best is None else min(best, ms)
            return best, r

        tgf = time_grid(NFE, SWAY)
        condp = torch.nn.functional.pad(cond, (0, 0, 0, N_TOT - F_REF))
        hoist_ms, _ = timed_ms(lambda: eng.prepare(condp, cm, text, tgf.numpy(), cond_frames=F_REF, cfg_strength=CFG))
        loop_ms, (o_, _y) = timed_ms(lambda: eng.solve(y0), reps=2)
        mel_in = o_[:, F_REF - 1:, :].permute(0, 2, 1).contiguous()
        voc_ms, wav_ = timed_ms(lambda: vocoder.decode(mel_in))
        d2h_ms, _ = timed_ms(lambda: host_wav.copy_(wav_, non_blocking=True))
        tot = hoist_ms + loop_ms + voc_ms + d2h_ms
        result["hoist_ms"] = hoist_ms
        result["phase_ms"] = {"hoists": hoist_ms, "step_loop": loop_ms, "vocoder": voc_ms, "d2h": d2h_ms, "sum_serial": tot,
                              "note": "one utterance batch, phases serialised and timed with stream events after the timed region; "
                                      "with --overlap 1 vocoder + d2h run under the next utterance's step loop"}
        result["phase_share"] = {"hoists": hoist_ms / tot, "step_loop": loop_ms / tot, "vocoder": voc_ms / tot, "d2h": d2h_ms / tot}
        # kernel_time_share covers the step loop; rescaled to the utterance and joined by the hoist / vocoder phases
        result["kernel_time_share_utterance"] = dict({k: round(v * loop_ms / tot, 4) for k, v in result["kernel_time_share"].items()},
                                                     hoists=round(hoist_ms / tot, 4), vocoder=round(voc_ms / tot, 4), d2h=round(d2h_ms / tot, 4))
        wbytes = 4 * sum(int(np.prod(np.shape(v))) for v in vsd_host.values())
        voc_bytes = B * (400.0 * L_GEN + 1024.0 * (L_GEN - 1)) + wbytes      # weights + mel in + wav out
        result["roofline_vocoder"] = {"bound": "hbm", "achieved": voc_bytes / (voc_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                      "frac": voc_bytes / (voc_ms * 1e-3) / 8e12, "algorithmic_bytes": voc_bytes, "decode_ms": voc_ms,
                                      "frames": L_GEN, "traffic": None,
                                      "note": "launch-/latency-bound at batch 1: ~60 small fp32 launches for 25 GFLOP"}
        result["clock_power"] = None if a.no_clock_power else clock_power(step)
        if result["clock_power"]:
            # the MFMA peak the roofline prices against is the 2.4 GHz figure; what the package was clocked to deliver while this ran
            result["roofline"]["peak_at_measured_clock"] = result["roofline"]["peak"] * result["clock_power"]["sclk_mhz"] / 2400.0
        if not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd_host, vsd_host, arch, w)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
