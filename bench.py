#!/usr/bin/env python
"""Headline benchmark of the MI355X-native LEMAS-TTS acoustic path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): audio-seconds/sec @24 kHz, NFE=32, CFG on.  A "step" is ONE pass of the hot path over one
batch of synthetic input = one utterance batch: per-utterance hoists (text embedding, conditioning projection,
time/AdaLN tables -- recomputed every utterance, nothing is cached across steps) + 32 Euler steps of the CFG-folded
DiT + Vocos decode of the generated frames + device->host copy of the waveform.  Inputs (reference mel, token ids,
noise) are resident in HBM when the timed region starts; model load is outside it.

Workload at every N: BASELINE configs[1] -- multilingual_grl, batch 1, 10 s reference + 10 s target (F=938, N=1875,
L_gen=938 -> 9.995 s of audio per step), NFE 32, cfg 2.0, sway coef 5 (capped 3.486), bf16 MFMA operands with fp32
accumulate/residual/ODE state.  Multi-GPU: utterance-level data parallelism, one process per GPU, weights generated
on rank 0 and broadcast over RCCL/xGMI, no collective in the step loop, fixed work per GPU (weak scaling).

The JSON line also carries
  roofline     -- dominant kernel (the bf16 MFMA GEMM family; the class with the largest total time), algorithmic
                  FLOPs per launch / average launch duration measured with HIP events on the launch stream in a
                  separate short eager pass (events cannot sit inside the replayed hipGraph), vs 2.5 PFLOP/s dense bf16;
  cpu_baseline -- the fp32 oracle (oracle/lemas_oracle.py, a port of the reference's path) timed on this box's host
                  cores on a bounded sample (1 of the 32 Euler steps at full N, scaled x32, + the full vocoder).
"""
from __future__ import annotations

import argparse
import json
import os
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
F_REF, N_TOT, NFE, CFG, SWAY = 938, 1875, 32, 2.0, 5
HOP, SR = 256, 24000
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_FP8_PEAK_TFLOPS = 5000.0


def fwd_flops(B: int, N: int) -> float:
    """FLOPs of one DiT forward with the AdaLN/time GEMVs hoisted."""
    return B * (378_888_192.0 * N + 90_112.0 * N * N)


def class_flops(cls: str, rows: int, n: int, bb: int, d: int = 1024, ff: int = 2048, heads: int = 16) -> float:
    """Algorithmic FLOPs of one launch of a step-loop kernel class (rows = BB*N real frames, not the padded row space)."""
    return {"gemm_qk_rope": 2.0 * rows * 2 * d * d, "gemm_v_t": 2.0 * rows * d * d, "gemm_attn_out": 2.0 * rows * d * d,
            "gemm_ff1_gelu": 2.0 * rows * ff * d, "gemm_ff2": 2.0 * rows * d * ff,
            "attention": 4.0 * n * n * 64 * bb * heads}[cls]


def build_inputs(rank_seed: int, device):
    cond = torch.from_numpy(synth.synth_cond_mel(1234 + rank_seed, F_REF))[None]
    nt = round(N_TOT * 0.17)
    text = torch.from_numpy(synth.synth_tokens(1234 + rank_seed, nt, VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(1234 + rank_seed, N_TOT))[None]
    return cond.to(device), text.to(device), y0.to(device)


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


def cpu_baseline(sd, vsd, arch):
    """Oracle on host cores, bounded sample of the same workload."""
    from oracle import lemas_oracle as O  # checker / baseline only
    cores = usable_cores()
    torch.set_num_threads(cores)
    cond = torch.from_numpy(synth.synth_cond_mel(1234, F_REF))[None]
    text = torch.from_numpy(synth.synth_tokens(1234, round(N_TOT * 0.17), VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(1234, N_TOT))[None]
    cfm = O.OracleCFM(sd, arch)
    sub = 1
    tg = O.time_grid(NFE, SWAY)[: sub + 1]
    t0 = time.perf_counter()
    out, _ = cfm.sample(cond, text, N_TOT, y0=y0, steps=sub, cfg_strength=CFG, sway_sampling_coef=SWAY, t_grid=tg)
    t_steps = time.perf_counter() - t0
    mel = out[:, F_REF - 1:, :].permute(0, 2, 1)
    t0 = time.perf_counter()
    O.OracleVocos(vsd).decode(mel)
    t_voc = time.perf_counter() - t0
    audio_s = HOP * (mel.shape[-1] - 1) / SR
    est = t_steps * (NFE / sub) + t_voc
    return {"value": audio_s / est, "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "sample": f"{sub} of {NFE} Euler steps at N={N_TOT} ({t_steps:.1f} s) scaled x{NFE // sub} + full Vocos decode "
                      f"({t_voc:.2f} s); fp32 torch, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--depth", type=int, default=22, help="DiT depth (22 = the shipped model; smaller only for debugging)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", type=int, default=0, help="1 = block GEMMs on the fp8-e4m3 (MXFP8) path of BASELINE config 5; "
                    "NOT the headline configuration (configs[1] is bf16): the line is then labelled dtype fp8")
    ap.add_argument("--dual", type=int, default=1, help="1 = CFG branches as two concurrent lanes (default), 0 = one stream")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # LEMAS_DIST_BACKEND=gloo + LEMAS_SHARE_GPU=1 exist only to rehearse the N>1 code path on a 1-GPU box
        backend = os.environ.get("LEMAS_DIST_BACKEND", "nccl")
        if os.environ.get("LEMAS_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU path)"
    device = torch.device(f"cuda:{local_rank}")
    comm_device = device if (world == 1 or os.environ.get("LEMAS_DIST_BACKEND", "nccl") == "nccl") else torch.device("cpu")
    torch.cuda.set_device(device)

    from lemas_tts_amd.engine import VocosEngine  # noqa: E402
    from lemas_tts_amd.model.cfm import CFM, time_grid  # noqa: E402
    from lemas_tts_amd.parallel import broadcast_state_dict  # noqa: E402

    arch = DiTArch(depth=a.depth)
    # weights: generated on rank 0, broadcast over RCCL/xGMI to the other ranks (one flat fp32 blob)
    sd = synth.synth_cfm_state_dict(arch, VOCAB, 1234) if rank == 0 else None
    vsd = synth.synth_vocos_state_dict(1234) if rank == 0 else None
    if world > 1:
        sd = broadcast_state_dict(sd, arch, VOCAB, comm_device, dist)
        vsd = broadcast_state_dict(vsd, None, None, comm_device, dist, vocos=True)
    model = CFM(arch, VOCAB, sd, device=device)
    model.engine.set_option("dual", a.dual)
    model.engine.set_option("fp8", a.fp8)
    if os.environ.get("LEMAS_QKV_FUSED") is not None:
        model.engine.set_option("qkv_fused", int(os.environ["LEMAS_QKV_FUSED"]))
    model.engine.set_option("table_cache", 0)      # hoists are redone for every utterance: nothing cached across steps
    vocoder = VocosEngine(vsd, device=device)
    cond, text, y0 = build_inputs(rank, device)
    L_GEN = N_TOT - F_REF + 1
    host_wav = torch.empty((1, HOP * (L_GEN - 1)), dtype=torch.float32).pin_memory()

    def step():
        out, _ = model.sample(cond, text, N_TOT, steps=NFE, cfg_strength=CFG, sway_sampling_coef=SWAY, y0=y0, use_acc_grl=False)
        # the driver vocodes generated[:, nw // 256:] = frames F-1.. (utils_infer.py:520,546): L_gen = N - F + 1
        wav = vocoder.decode(out[:, F_REF - 1:, :].permute(0, 2, 1))
        host_wav.copy_(wav, non_blocking=True)
        return wav

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
    if dist:
        tmax = torch.tensor([elapsed], device=comm_device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert np.isfinite(host_wav.numpy()).all()

    result = None
    if rank == 0:
        audio_per_step = HOP * (L_GEN - 1) / SR
        value = world * a.steps * audio_per_step / elapsed
        flops_step = 2 * NFE * fwd_flops(1, N_TOT) * (a.depth / 22.0)
        result = {
            "metric": "audio-seconds/sec @24kHz (NFE=32, CFG on)", "value": value, "unit": "audio-seconds/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8" if a.fp8 else "bf16", "data": "synthetic",
            "rtf": elapsed / (a.steps * audio_per_step),
            "config": {"workload": "BASELINE configs[1]: multilingual_grl, batch 1, 10 s ref + 10 s target "
                                   f"(F={F_REF}, N={N_TOT}), NFE={NFE}, CFG={CFG}, sway coef {SWAY} (capped), "
                                   + ("fp8-e4m3 (MXFP8) GEMM operands, bf16 attention" if a.fp8 else "bf16 MFMA operands")
                                   + " / fp32 state, Vocos decode + D2H included",
                       "utterances_per_gpu_per_step": 1, "audio_seconds_per_step": audio_per_step,
                       "parallelism": f"dp{world} (utterance sharding, RCCL weight broadcast, no step-loop collectives)",
                       "depth": a.depth, "weights": "synthetic N(0,0.02^2), seed 1234"},
            "path_tflops": flops_step / (elapsed / a.steps) / 1e12,
        }

    # ---- roofline of the dominant kernel: short eager pass with per-launch HIP events (rank 0, N == 1 only)
    if rank == 0 and world == 1:
        eng = model.engine
        eng.set_option("profile", 1)
        sub = 4
        tg = time_grid(NFE, SWAY)[: sub + 1]
        cm = torch.zeros(1, N_TOT, dtype=torch.bool)
        cm[:, :F_REF] = True
        eng.prepare(torch.nn.functional.pad(cond, (0, 0, 0, N_TOT - F_REF)), cm, text, tg.numpy(), cond_frames=F_REF, cfg_strength=CFG)
        eng.solve(y0, want_out=False)
        prof = eng.profile_read()
        eng.set_option("profile", 0)
        mm = {k: v for k, v in prof.items() if k in ("gemm_qk_rope", "gemm_v_t", "gemm_attn_out", "gemm_ff1_gelu", "gemm_ff2", "attention")}
        dom = max(mm, key=lambda k: mm[k][0])      # dominant kernel = largest total time in the step loop
        ms, cnt = mm[dom]
        avg_us = 1e3 * ms / max(cnt, 1)
        lanes = 2 if a.dual else 1                 # dual: each launch covers one CFG branch (B rows of the 2B)
        rows, bb = 2 * N_TOT // lanes, 2 // lanes
        fl = class_flops(dom, rows, N_TOT, bb)
        ach = fl / (avg_us * 1e-6) / 1e12
        total_ms = sum(v[0] for v in prof.values())
        kname = "attn_fwd_splitkv_kernel" if dom == "attention" else f"gemm_bf16_kernel<{dom}>"
        traffic = None   # HBM-side bytes per launch from the committed PMC passes (rocprofv3 cannot run inside bench.py)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01o_traffic.json" if lanes == 2 else "r01_traffic.json")))
            traffic = tj[dom]["hbm_bytes_per_launch"]        # r01o: per-lane launches (one CFG branch); r01: both branches in one launch
        except (OSError, KeyError, ValueError):
            pass
        peak = MFMA_FP8_PEAK_TFLOPS if (a.fp8 and dom != "attention") else MFMA_BF16_PEAK_TFLOPS   # attention stays bf16
        result["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": peak,
                              "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                              "avg_launch_us": avg_us, "launches": int(cnt), "flops_per_launch": fl,
                              "launch_shape": f"{bb} x {N_TOT} frames x 16 heads per launch ({lanes} concurrent lane(s))"}
        result["kernel_tflops"] = {k: round(class_flops(k, rows, N_TOT, bb) / (1e3 * v[0] / max(v[1], 1) * 1e-6) / 1e12, 1)
                                   for k, v in mm.items()}
        result["kernel_time_share"] = {k: round(v[0] / total_ms, 4) for k, v in prof.items()}
        result["kernel_avg_us"] = {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in prof.items()}
        if not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd, vsd, arch)
    if rank == 0:
        print(json.dumps(result))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
