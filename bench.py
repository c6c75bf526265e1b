#!/usr/bin/env python
"""Headline benchmark of the MI355X-native LEMAS-TTS acoustic path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload configs1|configs2|configs3|configs4|short] [--job configs3_full]

With ``--gpus N`` (N > 1) and no launcher environment, bench.py starts the N ranks ITSELF (one process per GPU, the reference's
own multi-GPU precedent: uvr5/multiprocess_cuda_infer.py:404-420); started under ``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`` it is one of the N ranks.  Either way the
world size must equal ``--gpus``.

Metric (BASELINE.json): audio-seconds/sec @24 kHz, NFE=32, CFG on.  A "step" is ONE pass of the hot path over one batch of
synthetic input = one utterance batch per GPU: per-utterance hoists (text embedding, conditioning projection, time/AdaLN
tables -- recomputed every utterance, nothing is cached across steps) + the Euler steps of the CFG-folded DiT + Vocos decode
of the generated frames + device->host copy of the waveform.  Inputs (reference mel, token ids, noise) are resident in HBM
when the timed region starts; model load is outside it.

Workloads (per GPU and step; fixed as N grows = weak scaling):
  configs1  BASELINE configs[1] (the headline): multilingual_grl, batch 1, 10 s reference + 10 s target (F=938, N=1875)
  configs2  BASELINE configs[2]: multilingual_prosody, batch 8 of MIXED lengths (prompts 375..938 frames, durations 900..1900: ragged
            `lens` / masks), prosody conditioning, sway; every utterance vocoded on its own.  The line reports the padded-row waste
            (real frames vs the rows the batched launches compute)
  configs3  BASELINE configs[3]'s per-GPU share: 8 utterances of 4 s reference + 8 s target (F=375, N=1125) as one batch
            (64 utterances over 8 GPUs)
  configs4  BASELINE configs[4]: speech-edit infill of a 30 s source (N=2814), 3 edit spans, NFE 48, sway 3, fp8 MFMA weights (the
            workload's own precision: --fp8 defaults to 1 here), the WHOLE utterance vocoded (speech_edit_multilingual.py:193-198)
  short     one short utterance, 4 s + 4 s (F=375, N=750): what the entry script mostly sees
All: cfg 2.0, bf16 MFMA operands with fp32 accumulate / residual / ODE state unless fp8 is on; NFE 32 and sway coef 5 (capped
3.486) except configs4.
Multi-GPU: utterance-level data parallelism, weights generated on rank 0 and broadcast over RCCL/xGMI into device memory
(loaded device-to-device on every rank), no collective in the step loop.  ``--job configs3_full`` runs the SHARDED JOB instead of
the weak-scaling step: rank 0 owns BASELINE configs[3]'s 64 utterances, deals them (lemas_tts_amd.parallel.shard_utterances), every
rank samples + vocodes its shard, the waveforms come back to rank 0's host, and the wall time of the job is reported (see run_job).

CORRECTNESS inside the run: the mel the timed region produced last is compared with the committed output of the REFERENCE
itself on the same inputs (tests/golden/configs*.npz: 22 blocks, full NFE; the short workload against configs0_nfe16.npz, the same
utterance over 16 steps, in one extra untimed solve; made by oracle/gen_golden.py --full-size); the run FAILS above mel-MSE 1e-4
and the value is reported as ``mel_mse_vs_reference``.

Pipelining (``--overlap 1``, the default): the Vocos decode + D2H of utterance i run on a side stream under the step loop of
utterance i+1 (the step loop itself still holds ONE utterance batch at a time: the B = 1 definition of configs[1] is unchanged;
every waveform is on the host when the timed region ends).  ``--overlap 0`` serialises them as the reference's loop does.

The JSON line also carries
  roofline     -- the dominant kernel BY SYMBOL (what rocprofv3 --stats lists; out-proj and FF2 share one instantiation):
                  algorithmic FLOPs per launch / average launch duration, measured live with HIP event pairs stamped by the
                  dispatch itself (hipExtLaunchKernelGGL) in a short eager pass with the timed region's launch shapes, vs
                  2.5 PFLOP/s dense bf16, plus what the committed PMC passes say the kernel is bound BY (profiles/r04/r04_kernel_bounds.json:
                  matrix-pipe busy share, waves parked / stalled, L2 hit rate, fabric bytes); ``roofline_kernels`` = the same for every
                  big launch; ``roofline_gemm_family`` = all block GEMMs together, ``path_frac`` = whole path;
  roofline_vocoder -- the vocoder phase: algorithmic fp32 FLOPs of one decode over its measured duration vs the 157.3 TFLOP/s exact-fp32
                  MFMA peak (what bounds it); ``hbm_view`` keeps SURVEY.md 8d's byte pricing (fp32 weights + 400 L + 1024 (L-1) vs 8 TB/s); ``phase_ms`` = hoists / step loop / vocoder / D2H of one utterance, measured
                  serially with stream events after the timed region;
  cpu_baseline -- the fp32 oracle (oracle/lemas_oracle.py, a port of the reference's path) timed on this box's host
                  cores on a bounded sample: the FIRST Euler step (which also builds both branches' text embedding, cached afterwards)
                  and a SECOND, warm one are timed apart; estimate = first + 31 x warm + the full vocoder.

Measurement switches (none changes what the timed region computes): ``--no-phases`` skips the serial hoists / step loop / vocoder / D2H
timing after the timed region (profiler passes with counters on the batched workloads: DESIGN.md section 8, profiler note),
``--no-clock-power`` the rocm-smi sampling pass, ``--no-cpu-baseline`` the oracle leg; ``--graph 0`` / ``--vocoder-graph 0`` launch eagerly,
``--skip-dead 1|2`` / ``--skip-masked 0`` select what a ragged batch's padding blocks cost (the attention half of a block
skips them by default, exactly; the FF half on request), ``--xcd-runs 1`` is round 3's GEMM tile order, ``--attn-variant`` / ``--ln-fused`` / ``--ln-fold`` / ``--dual 0`` select the engine's
non-default kernels and schedules (the A/Bs of DESIGN.md section 8).
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

# vocode: "generated" = frames F-1.. of every utterance (utils_infer.py:520,546) | "each" = the same per utterance of a ragged batch |
# "whole" = every frame (speech_edit_multilingual.py:193-198)
WORKLOADS = {
    "configs1": dict(B=1, F=938, N=1875, golden="configs1_nfe32.npz", golden_steps=32, nfe=32, sway=5, wseed=1234, prosody=False, fp8=0, vocode="generated",
                     desc="BASELINE configs[1]: multilingual_grl, batch 1, 10 s ref + 10 s target"),
    "configs2": dict(B=8, F=938, N=1900, golden="configs2_prosody_b8.npz", golden_steps=32, nfe=32, sway=5, wseed=1235, prosody=True, fp8=0, vocode="each",
                     desc="BASELINE configs[2]: multilingual_prosody, batch 8 of mixed lengths (prompts 375..938, durations 900..1900 frames), "
                          "prosody conditioning, sway"),
    "configs3": dict(B=8, F=375, N=1125, golden="configs3_share_nfe32.npz", golden_steps=32, nfe=32, sway=5, wseed=1234, prosody=False, fp8=0, vocode="generated",
                     desc="BASELINE configs[3] per-GPU share: multilingual_grl, 8 utterances of 4 s ref + 8 s target as one batch"),
    "configs4": dict(B=1, F=2813, N=2814, golden="configs4_edit_nfe48.npz", golden_steps=48, nfe=48, sway=3.0, wseed=1234, prosody=False, fp8=1, vocode="whole",
                     desc="BASELINE configs[4]: speech-edit infill of a 30 s source, 3 edit spans (7.5 s regenerated), NFE 48, sway 3, fp8 MFMA weights, "
                          "whole utterance vocoded"),
    "short": dict(B=1, F=375, N=750, golden="configs0_nfe16.npz", golden_steps=16, nfe=32, sway=5, wseed=1234, prosody=False, fp8=0, vocode="generated",
                  desc="one short utterance: multilingual_grl, batch 1, 4 s ref + 4 s target"),
}


def fwd_flops(B: int, N: int) -> float:
    """FLOPs of one DiT forward with the AdaLN/time GEMVs hoisted."""
    return B * (378_888_192.0 * N + 90_112.0 * N * N)


def class_flops(cls: str, rows: float, nsq: float, d: int = 1024, ff: int = 2048, heads: int = 16) -> float:
    """Algorithmic FLOPs of one launch of a step-loop kernel class: rows = REAL frames the launch covers (not the padded row space),
    nsq = sum over its samples of (real frames)^2."""
    return {"gemm_qkv_fused": 2.0 * rows * 3 * d * d, "gemm_qk_rope": 2.0 * rows * 2 * d * d, "gemm_v_t": 2.0 * rows * d * d,
            "gemm_attn_out": 2.0 * rows * d * d, "gemm_ff1_gelu": 2.0 * rows * ff * d, "gemm_ff2": 2.0 * rows * d * ff,
            "attention": 4.0 * nsq * 64 * heads}[cls]


def class_bytes(cls: str, rows: float, nsq_over_rows: float, d: int = 1024, ff: int = 2048) -> float:
    """Algorithmic HBM bytes of one launch (operands in + results out, each once; bf16 activations and weights, the
    fp32 residual stream read and written by the gate + residual launches)."""
    return {"gemm_qkv_fused": rows * d * 2 + 3 * d * d * 2 + rows * 3 * d * 2, "gemm_qk_rope": rows * d * 2 + 2 * d * d * 2 + rows * 2 * d * 2,
            "gemm_v_t": rows * d * 2 + d * d * 2 + rows * d * 2, "gemm_attn_out": rows * d * 2 + d * d * 2 + 2 * rows * d * 4,
            "gemm_ff1_gelu": rows * d * 2 + ff * d * 2 + rows * ff * 2, "gemm_ff2": rows * ff * 2 + ff * d * 2 + 2 * rows * d * 4,
            "attention": 4 * rows * d * 2}[cls]


def latency_roofline(kernel_avg_us: dict, pipes: dict, rows_lane: int, block_us_measured: float, lanes: int, clock_ghz: float):
    """The LATENCY roofline of one CFG lane's block at batch 1: the block is a dependent chain of seven launches (LN, QK+V, attention, out-proj,
    LN, FF1, FF2), so its floor is not FLOPs / peak but   sum over the launches of (ideal K loop + what a launch costs outside its loop) +
    boundaries.  Inputs: per K-tile pipe times and the prologue / epilogue / ramp + drain of the 128 x 128 tile from the committed ablation
    builds (`pipes`, profiles/r04/r04g_kloop_pipes.json), the launch boundary from MI355X_MICROARCH.md's price list ("boundary": 1.1-1.4 us inside a
    GEMM chain), and this run's own per-launch averages (eager pass) for the launches the model has no pipe figures for (LayerNorm, attention).
    An "ideal K loop" runs at the slower of its two equally loaded pipes -- 512 MFMA clocks per SIMD and 32 KB through the CU's 64 B/clk
    vector-memory path per 128 x 128 x 64 K-tile -- i.e. with the fragment reads and the LDS-DMA requests perfectly hidden."""
    t17 = pipes["tiles"]["17"]
    fixed = t17["outside_the_loop_us"]
    out_loop = fixed["prologue"] + fixed["epilogue"] + fixed["launch_ramp_and_drain"]
    boundary = 1.25
    tiles_m = (rows_lane + 127) // 128
    cus = 256 // lanes if lanes > 1 else 256                    # with two lanes on the chip a lane's launch has half of it
    def gemm(n, k, bm=128, bn=128):
        tiles = ((rows_lane + bm - 1) // bm) * (n // bn)
        rounds = -(-tiles // cus)
        clk = max(bm * bn / 32.0, 2.0 * (bm + bn))               # MFMA clocks per SIMD vs bytes / 64 B per clock, per K-tile
        return rounds * (k // 64) * clk / (clock_ghz * 1e3)
    chain = {
        "ln_mod_x2": 2 * kernel_avg_us.get("ln_mod", 5.0),       # measured (one round trip in, one write-through out: no pipe model)
        "qkv": gemm(3072, 1024, 256, 128) + out_loop,
        "attention": kernel_avg_us.get("attention", 22.0),       # measured: VALU (softmax) and MFMA about equally loaded, no K-loop model
        "out_proj": gemm(1024, 1024) + out_loop,
        "ff1": gemm(2048, 1024) + out_loop,
        "ff2": gemm(1024, 2048) + out_loop,
    }
    ideal = sum(chain.values()) + 7 * boundary
    alone = {k: kernel_avg_us.get(k) for k in ("ln_mod", "gemm_qkv_fused", "attention", "gemm_attn_out", "gemm_ff1_gelu", "gemm_ff2")}
    have = all(v is not None for v in alone.values())
    alone_sum = (2 * alone["ln_mod"] + sum(v for k, v in alone.items() if k != "ln_mod") + 7 * boundary) if have else None
    return {"what": "per DiT block and CFG lane, dependent chain of 7 launches: sum of (ideal K loop at the measured pipe rates + measured prologue / "
                    "epilogue / ramp + drain) + 7 launch boundaries; LayerNorm and attention at their measured stand-alone times",
            "chain_ideal_us": round(ideal, 1), "chain_ideal_terms_us": {k: round(v, 2) for k, v in chain.items()},
            "chain_of_measured_launches_us": round(alone_sum, 1) if alone_sum else None,
            "block_measured_us": round(block_us_measured, 1),
            "frac": round(ideal / block_us_measured, 3),
            "inputs": {"outside_the_loop_us": fixed, "boundary_us": boundary, "boundary_source": "MI355X_MICROARCH.md price list, row 'boundary' (1.1-1.4 inside a GEMM chain)",
                       "cus_per_lane": cus, "clock_ghz": clock_ghz, "tiles_m": tiles_m, "pipes_source": pipes.get("_source")},
            "reading": "block_measured_us is both lanes' blocks overlapped on one chip (one block of each lane per block_measured_us); a lane alone on half "
                       "the chip cannot finish its block faster than chain_ideal_us, and the MFMA roofline of the same block (FLOPs / peak) is "
                       "several times lower than either"}


# kernel classes of the profile pass -> the symbol rocprofv3 lists them under (out-proj and FF2 are ONE instantiation)
SYMBOL = {"gemm_qkv_fused": "gemm_qkv_fused_kernel", "gemm_qk_rope": "gemm_bf16_kernel<EPI_QK_ROPE>", "gemm_v_t": "gemm_bf16_kernel<EPI_V_T>",
          "gemm_attn_out": "gemm_bf16_kernel<EPI_GATE_RES>", "gemm_ff2": "gemm_bf16_kernel<EPI_GATE_RES>",
          "gemm_ff1_gelu": "gemm_bf16_kernel<EPI_BIAS_GELU>", "attention": "attn_fwd_splitkv_kernel"}


def build_inputs(w: dict, rank: int, device):
    """rank 0 runs the utterance(s) of the committed reference fixture (so the result can be checked); the others seeded ones
    (equal-length workloads; tools/e2e_ab.py uses this form)"""
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


def build_case(w: dict, rank: int, device):
    """Everything one CFM.sample call of the workload takes.  The ragged / edit workloads (configs2, configs4) run the committed reference
    fixture's own inputs on every rank (weak scaling: the same work everywhere); the equal-length ones as build_inputs."""
    if w["vocode"] == "generated":
        cond, text, y0, fx = build_inputs(w, rank, device)
        B = w["B"]
        return dict(cond=cond, text=text, y0=y0, duration=w["N"], lens=None, kw={}, lens_list=[w["F"]] * B, dur_list=[w["N"]] * B, fx=fx)
    fx = synth.expand_reference_fixture(dict(np.load(os.path.join(GOLDEN, w["golden"]))))
    B = int(fx["B"])
    kw = {}
    if "edit_mask" in fx:
        kw["edit_mask"] = torch.from_numpy(fx["edit_mask"])
    if "prosody_embeds" in fx:
        kw["prosody_embeds"] = torch.from_numpy(fx["prosody_embeds"]).to(device)
    dur = [int(v) for v in fx["duration"]]
    return dict(cond=torch.from_numpy(fx["cond"]).to(device), text=torch.from_numpy(fx["text"]).to(device), y0=torch.from_numpy(fx["y0"]).to(device),
                duration=dur[0] if B == 1 else torch.from_numpy(fx["duration"]), lens=torch.from_numpy(fx["lens"]), kw=kw,
                lens_list=[int(v) for v in fx["lens"]], dur_list=dur, fx=(fx if rank == 0 else None))


def vocode_segments(w: dict, case: dict, out_frames: int):
    """(sample, first frame, end frame) of every waveform the workload produces; a decode of L frames yields HOP (L - 1) samples"""
    if w["vocode"] == "whole":
        return [(b, 0, out_frames) for b in range(w["B"])]
    return [(b, case["lens_list"][b] - 1, min(case["dur_list"][b], out_frames)) for b in range(w["B"])]


def gen_mse(out: np.ndarray, ref: np.ndarray, fx: dict) -> float:
    """mel-MSE over the generated (non-conditioning) frames of every sample, as SURVEY.md 8d defines it (tests/test_gpu_00_sample.py)"""
    se, cnt = 0.0, 0
    for b in range(int(fx["B"])):
        L, D = int(fx["lens"][b]), int(fx["duration"][b])
        keep = np.ones(out.shape[1], bool)
        keep[:L] = False
        if "edit_mask" in fx:
            keep = ~(np.pad(fx["edit_mask"][b], (0, out.shape[1] - fx["edit_mask"].shape[1])) & (np.arange(out.shape[1]) < L))
        keep &= np.arange(out.shape[1]) < D
        d = out[b, keep].astype(np.float64) - ref[b, keep].astype(np.float64)
        se += float((d ** 2).sum())
        cnt += d.size
    return se / max(cnt, 1)


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
    """Oracle on host cores, bounded sample of the same workload: ONE utterance of the workload's prompt / target size, the first and a
    second (warm) Euler step timed apart -- the first also builds both CFG branches' text embedding, which the reference caches for the
    rest of the solve (dit.py:212-220) -- and the full vocoder."""
    from oracle import lemas_oracle as O  # checker / baseline only
    cores = usable_cores()
    torch.set_num_threads(cores)
    F, N, nfe = w["F"], w["N"], w["nfe"]
    cond = torch.from_numpy(synth.synth_cond_mel(1234, F))[None]
    text = torch.from_numpy(synth.synth_tokens(1234, round(N * 0.17), VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(1234, N))[None]
    if "prosody_to_mel.weight" in sd:      # the prosody workload's weights: time the same DiT without the (B x 512) conditioning GEMVs
        sd = {k: v for k, v in sd.items() if not (k.startswith("prosody_to_mel.") or ".prosody_text_proj." in k)}
    cfm = O.OracleCFM(sd, arch)
    tg = O.time_grid(nfe, w["sway"])
    t0 = time.perf_counter()
    cfm.sample(cond, text, N, y0=y0, steps=1, cfg_strength=CFG, sway_sampling_coef=w["sway"], t_grid=tg[:2])
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    out, _ = cfm.sample(cond, text, N, y0=y0, steps=2, cfg_strength=CFG, sway_sampling_coef=w["sway"], t_grid=tg[:3])
    t_two = time.perf_counter() - t0
    t_warm = max(t_two - t_first, 1e-3)
    lo = 0 if w["vocode"] == "whole" else F - 1
    mel = out[:, lo:, :].permute(0, 2, 1)
    t0 = time.perf_counter()
    O.OracleVocos(vsd).decode(mel)
    t_voc = time.perf_counter() - t0
    audio_s = HOP * (mel.shape[-1] - 1) / SR
    est = t_first + (nfe - 1) * t_warm + t_voc
    return {"value": audio_s / est, "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "first_step_s": t_first, "warm_step_s": t_warm, "vocoder_s": t_voc,
            "sample": f"one utterance of the workload's size (F={F}, N={N}): Euler step 1 of {nfe} ({t_first:.1f} s, incl. both branches' text "
                      f"embedding) and a warm step 2 ({t_warm:.1f} s, from a 2-step solve) timed apart, estimate = first + {nfe - 1} x warm + full Vocos "
                      f"decode ({t_voc:.2f} s); fp32 torch, {cores} threads"}


def clock_power(step_fn, seconds: float = 4.0):
    """Engine clock and package power WHILE the timed loop's work runs (an untimed repeat of it), read from ``rocm-smi`` in a second
    process: the board manages the clock against a 1400 W package limit, and the solve runs ~9 % under the 2.4 GHz the MFMA peak is
    quoted at (profiles/r03/r03_power_clocks.txt).  None when rocm-smi is missing or prints something else."""
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


def run_job(a, w, model, vocoder, rank, world, dist, comm_device, device, bcast):
    """BASELINE configs[3] as a JOB (utterance shards scattered from the host, waveforms returned to it; precedent
    uvr5/multiprocess_cuda_infer.py:404-420): rank 0 owns the 64 utterances (4 s prompt + 8 s target each), deals them longest-first
    (lemas_tts_amd.parallel.shard_utterances), sends every rank its shard's inputs, each rank runs its shard as ONE CFM.sample batch
    + one Vocos decode and the waveforms travel back to rank 0's HOST memory.  Reported: wall time of the job on rank 0 (inputs
    leaving its host to last waveform back), per-rank compute, scatter and gather times.  No collective inside the step loop."""
    from lemas_tts_amd.parallel import shard_utterances
    U, F_, N = 64, w["F"], w["N"]
    nt = round(N * 0.17)
    L = N - F_ + 1
    nfe, sway = w["nfe"], w["sway"]
    shards = shard_utterances([N] * U, world)
    mine = shards[rank]
    per = len(mine)
    assert sorted(i for sh in shards for i in sh) == list(range(U)) and all(len(sh) == U // world for sh in shards), "64 utterances deal evenly"

    def make(i):          # utterance i of the job (seeded: every rank COULD build it, only rank 0 does)
        return (torch.from_numpy(synth.synth_cond_mel(5000 + i, F_)), torch.from_numpy(synth.synth_tokens(5000 + i, nt, VOCAB)),
                torch.from_numpy(synth.synth_noise(5000 + i, N)))
    job = [make(i) for i in range(U)] if rank == 0 else None

    from lemas_tts_amd.parallel import gather_to_rank0, scatter_from_rank0
    MB = w["B"]           # utterances per CFM.sample batch: configs[3]'s per-GPU batch of 8; a rank with a longer shard runs it in turns

    def pack(idx):        # a shard as two flat tensors: [cond | y0] fp32, token ids int64
        return [torch.cat([torch.stack([job[i][0] for i in idx]).reshape(-1), torch.stack([job[i][2] for i in idx]).reshape(-1)]),
                torch.stack([job[i][1] for i in idx]).reshape(-1)]
    packed = [pack(shards[r]) for r in range(world)] if rank == 0 else None        # rank 0 owns the job, already laid out per shard
    like = [torch.empty(per * (F_ + N) * 100, dtype=torch.float32), torch.empty(per * nt, dtype=torch.int64)]

    def once():
        t = {}
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # ---- scatter: rank 0 sends each rank its shard
        fbuf, tbuf = scatter_from_rank0(packed, like, dist if world > 1 else None, comm_device)
        cond = fbuf[: per * F_ * 100].reshape(per, F_, 100).to(device)
        y0 = fbuf[per * F_ * 100:].reshape(per, N, 100).to(device)
        text = tbuf.reshape(per, nt).cpu()
        torch.cuda.synchronize()
        t["scatter_ms"] = 1e3 * (time.perf_counter() - t0)
        # ---- this rank's shard: batches of MB through the sampler and the vocoder
        t1 = time.perf_counter()
        wavs, out = [], None
        for c0 in range(0, per, MB):
            o, _ = model.sample(cond[c0:c0 + MB], text[c0:c0 + MB], N, steps=nfe, cfg_strength=CFG, sway_sampling_coef=sway, y0=y0[c0:c0 + MB],
                                use_acc_grl=False)
            wavs.append(vocoder.decode(o[:, F_ - 1:, :].permute(0, 2, 1)))
            out = o if out is None else out
        wav = torch.cat(wavs)
        torch.cuda.synchronize()
        t["compute_ms"] = 1e3 * (time.perf_counter() - t1)
        # ---- gather: waveforms back to rank 0's host
        t2 = time.perf_counter()
        got = gather_to_rank0(wav, dist if world > 1 else None, comm_device)
        host = None
        if rank == 0:
            host = torch.empty((U, HOP * (L - 1)), dtype=torch.float32)
            for r in range(world):
                host[torch.tensor(shards[r])] = got[r]
        torch.cuda.synchronize()
        t["gather_ms"] = 1e3 * (time.perf_counter() - t2)
        t["job_ms"] = 1e3 * (time.perf_counter() - t0)
        return t, host, out

    for _ in range(max(a.warmup, 1)):
        once()
    runs = [once() for _ in range(max(a.steps, 1))]
    times = [r[0] for r in runs]
    host = runs[-1][1]
    mine_t = torch.tensor([[t["scatter_ms"], t["compute_ms"], t["gather_ms"], t["job_ms"]] for t in times], dtype=torch.float64).mean(0)
    every = [mine_t]
    if dist and world > 1:
        every = [torch.zeros_like(mine_t, device=comm_device) for _ in range(world)]
        dist.all_gather(every, mine_t.to(comm_device))
        every = [e.cpu() for e in every]
    if rank != 0:
        return None
    assert host is not None and np.isfinite(host.numpy()).all() and float(host.abs().amax()) > 0
    # parity inside the run: one utterance of rank 0's shard alone gives the bits it gave inside the batch (what makes the deal irrelevant)
    i0 = shards[0][0]
    c1, t1_, y1 = make(i0)
    one, _ = model.sample(c1[None].to(device), t1_[None], N, steps=nfe, cfg_strength=CFG, sway_sampling_coef=sway, y0=y1[None].to(device), use_acc_grl=False)
    same = bool(torch.equal(one[0].cpu(), runs[-1][2][0].cpu()))
    assert same, "bench.py --job: an utterance inside its rank's batch differs from the same utterance alone"
    job_ms = float(every[0][3])
    audio = U * HOP * (L - 1) / SR
    return {"metric": f"audio-seconds/sec @24kHz (NFE={nfe}, CFG on): sharded JOB", "value": audio / (job_ms * 1e-3), "unit": "audio-seconds/sec",
            "n_gpus": world, "steps": max(a.steps, 1), "warmup": max(a.warmup, 1), "ms_per_step": job_ms, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "fp8" if a.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3] as a job: {U} utterances of 4 s ref + 8 s target (F={F_}, N={N}) owned by rank 0, dealt to "
                                   f"{world} rank(s) ({U // world} per rank, run as batches of {MB}), NFE={nfe}, CFG={CFG}, waveforms gathered on rank 0's host",
                       "workload_key": "configs3_full", "utterances_total": U, "utterances_per_rank": U // world, "utterances_per_batch": MB,
                       "batches_per_rank": (U // world + MB - 1) // MB,
                       "parallelism": f"dp{world}: scatter inputs -> per-rank CFM.sample batch + Vocos decode -> gather waveforms; no step-loop collective",
                       "audio_seconds_total": audio, "utterance_alone_equals_in_batch": same},
            "job_ms": job_ms,
            "per_rank_ms": {"scatter": [round(float(e[0]), 3) for e in every], "compute": [round(float(e[1]), 3) for e in every],
                            "gather": [round(float(e[2]), 3) for e in every], "job": [round(float(e[3]), 3) for e in every]},
            "weight_broadcast": bcast,
            "weight_broadcast_note": "fp32 as loaded (1.35 GB): the per-step AdaLN tables and every per-utterance hoist are computed from fp32 masters; "
                                     "only the block GEMM weights (0.37 GB of it in bf16 terms) could travel rounded -- a once-per-process saving of a few ms"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="configs1")
    ap.add_argument("--depth", type=int, default=22, help="DiT depth (22 = the shipped model; smaller only for debugging, no parity check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", type=int, default=-1, help="1 = block GEMMs on the fp8-e4m3 (MXFP8) path of BASELINE config 5, 0 = bf16; default: the "
                    "workload's own precision (bf16 everywhere but configs4).  The line is labelled with the dtype it ran")
    ap.add_argument("--job", choices=["configs3_full"], default=None, help="run the SHARDED JOB instead of the weak-scaling step: rank 0 owns "
                    "BASELINE configs[3]'s 64 utterances, deals them to the ranks, gathers the waveforms on its host (see run_job)")
    ap.add_argument("--attn-variant", type=int, default=-1, help="engine option attn_variant (-1 = engine default)")
    ap.add_argument("--no-phases", action="store_true", help="skip the serial hoists / step loop / vocoder / D2H timing after the timed region "
                    "(profiler passes: rocprofv3 --pmc aborts in that section on the batched workloads, DESIGN.md section 8)")
    ap.add_argument("--bcast-bf16", type=int, default=-1, help="N > 1: the DiT blocks' GEMM weights travel in bf16 (0.98 instead of 1.35 GB over xGMI; the bf16 step "
                    "loop's results are unchanged bit for bit).  -1 = on for the bf16 path, off for fp8 (its quantiser starts from the fp32 masters)")
    ap.add_argument("--xcd-runs", type=int, default=-1, help="engine measurement option xcd_runs (1 = round 3's GEMM tile order)")
    ap.add_argument("--graph", type=int, default=-1, help="engine option graph (-1 = engine default: one hipGraph launch per ODE step)")
    ap.add_argument("--vocoder-graph", type=int, default=-1, help="vocoder option graph (-1 = default: backbone + head replayed as one hipGraph)")
    ap.add_argument("--dual", type=int, default=1, help="1 = CFG branches as two concurrent lanes (default), 0 = one stream")
    ap.add_argument("--overlap", type=int, default=1, help="1 = Vocos decode + D2H of utterance i on a side stream under the step loop of "
                    "utterance i+1 (default), 0 = strictly serial")
    ap.add_argument("--attn-f8qk", type=int, default=-1, help="engine option attn_f8qk (-1 = engine default 1: on the fp8 path attention's QK^T runs on the "
                    "fp8 MFMA from MXFP8 q / k written by the QK epilogue; 0 = bf16 q / k; 2 = side-launch quantiser; +4 forces it on the bf16 path: a "
                    "measurement, not a BASELINE bf16 configuration)")
    ap.add_argument("--ln-fused", type=int, default=-1, help="engine option ln_fused (-1 = engine default)")
    ap.add_argument("--ln-fold", type=int, default=-1, help="engine option ln_fold (-1 = engine default)")
    ap.add_argument("--outlier-weights", type=int, default=0, help="1 = synthetic weights WITH outlier residual channels (1 %% of the channels x30 in every block's "
                    "residual-writing projections, synth.synth_cfm_state_dict(outlier=(0.01, 30))): what the fp8 path's outlier decomposition costs / buys; "
                    "the reference fixture does not apply to these weights (no parity check in the line)")
    ap.add_argument("--fp8-outlier-mode", type=int, default=-1, help="engine option fp8_outlier_mode (-1 = leave the default, 0: every block GEMM on bf16 operands when the guard trips; 1 = the mixed-precision decomposition, measurement builds only)")
    ap.add_argument("--skip-dead", type=int, default=-1, help="engine option skip_dead (-1 = engine default 0: every sample of a ragged batch runs at "
                    "the batch's pitch, as in the reference; 1 = the 128-row blocks that lie wholly in a sample's padding are left uncomputed: "
                    "+12.6 %% on configs2, the last ~30 frames of a sample then differ from the reference's by 5e-6 instead of 2e-6 mel-MSE)")
    ap.add_argument("--skip-masked", type=int, default=-1, help="engine option skip_masked (-1 = engine default 1: the attention half of every block "
                    "skips a ragged batch's padding blocks -- exact, the reference zeroes that half's output there; 0 = compute them, for A/B runs)")
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

    w = WORKLOADS["configs3" if a.job else a.workload]
    B, F_REF, N_TOT = w["B"], w["F"], w["N"]
    nfe, sway = w["nfe"], w["sway"]
    if a.fp8 < 0:
        a.fp8 = w["fp8"]
    arch = DiTArch(depth=a.depth)
    # weights: generated on rank 0, broadcast over RCCL/xGMI; every rank loads them device-to-device from the received buffer
    sd = synth.synth_cfm_state_dict(arch, VOCAB, w["wseed"], prosody=w["prosody"], outlier=(0.01, 30.0) if a.outlier_weights else None) if rank == 0 else None
    vsd = synth.synth_vocos_state_dict(1234) if rank == 0 else None
    sd_host, vsd_host = sd, vsd
    bcast = None
    if use_dist:
        t_b = time.perf_counter()
        from lemas_tts_amd.parallel import broadcast_bytes
        bf16_w = (not a.fp8 and a.ln_fold <= 0) if a.bcast_bf16 < 0 else bool(a.bcast_bf16)
        sd = broadcast_state_dict(sd, arch, VOCAB, comm_device, dist, prosody=w["prosody"], block_weights_bf16=bf16_w)
        vsd = broadcast_state_dict(vsd, None, None, comm_device, dist, vocos=True)
        nbytes = broadcast_bytes(arch, VOCAB, prosody=w["prosody"], block_weights_bf16=bf16_w) + broadcast_bytes(None, None, vocos=True)
        bcast = {"backend": backend, "bytes": nbytes, "seconds": time.perf_counter() - t_b, "block_gemm_weights": "bf16" if bf16_w else "fp32",
                 "on_device": bool(comm_device.type == "cuda"), "world": world}
    model = CFM(arch, VOCAB, sd, device=device, use_prosody_encoder=w["prosody"])
    if a.attn_variant >= 0:
        model.engine.set_option("attn_variant", a.attn_variant)
    if a.graph >= 0:
        model.engine.set_option("graph", a.graph)
    if a.xcd_runs >= 0:
        model.engine.set_option("xcd_runs", a.xcd_runs)
    model.engine.set_option("dual", a.dual)
    model.engine.set_option("fp8", a.fp8)
    if a.attn_f8qk >= 0:
        model.engine.set_option("attn_f8qk", a.attn_f8qk)
    if a.ln_fused >= 0:
        model.engine.set_option("ln_fused", a.ln_fused)
    if a.ln_fold >= 0:
        model.engine.set_option("ln_fold", a.ln_fold)
    if a.fp8_outlier_mode >= 0:
        model.engine.set_option("fp8_outlier_mode", a.fp8_outlier_mode)
    if a.skip_dead >= 0:
        model.engine.set_option("skip_dead", a.skip_dead)
    if a.skip_masked >= 0:
        model.engine.set_option("skip_masked", a.skip_masked)
    model.engine.set_option("table_cache", 0)      # hoists are redone for every utterance: nothing cached across steps
    vocoder = VocosEngine(vsd, device=device)
    if a.vocoder_graph >= 0:
        vocoder.set_option("graph", a.vocoder_graph)
    del sd, vsd                                     # the engines own their copies; the broadcast buffer can go
    if a.job:
        result = run_job(a, w, model, vocoder, rank, world, dist, comm_device, device, bcast)
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        sys.stdout.flush()
        if rank == 0:
            os.write(json_fd, (json.dumps(result) + "\n").encode())
        os.close(json_fd)
        return
    case = build_case(w, rank, device)
    fx = case["fx"]
    text = case["text"].cpu().pin_memory()          # token ids arrive from the host frontend (api.py:201-204): no D2H sync per utterance
    segs = vocode_segments(w, case, N_TOT)           # (sample, first frame, end frame) of every waveform one step produces
    host_wav = [torch.empty((1, HOP * (e - s - 1)), dtype=torch.float32).pin_memory() for _, s, e in segs]
    if w["vocode"] == "generated":                   # equal lengths: ONE decode of the whole batch, as before
        host_wav = [torch.empty((B, HOP * (segs[0][2] - segs[0][1] - 1)), dtype=torch.float32).pin_memory()]
    audio_per_step = sum(HOP * (e - s - 1) for _, s, e in segs) / SR
    last = {}
    side = torch.cuda.Stream(device) if a.overlap else None

    def vocode(out):
        # the driver vocodes generated[:, nw // 256:] = frames F-1.. (utils_infer.py:520,546): L_gen = N - F + 1; a ragged batch one
        # utterance at a time; the speech-edit script every frame (speech_edit_multilingual.py:193-198).  The slices go to the library as
        # strided views (VocosEngine.decode takes the frames-first layout in place): no torch copy kernel in between
        if w["vocode"] == "generated":
            wav = vocoder.decode(out[:, segs[0][1]:segs[0][2], :].permute(0, 2, 1))
            host_wav[0].copy_(wav, non_blocking=True)
            return wav
        wav = None
        for i, (b, s0, e0) in enumerate(segs):
            wav = vocoder.decode(out[b:b + 1, s0:e0, :].permute(0, 2, 1))
            host_wav[i].copy_(wav, non_blocking=True)
        return wav

    def sample(steps=nfe):
        return model.sample(case["cond"], text, case["duration"], lens=case["lens"], steps=steps, cfg_strength=CFG, sway_sampling_coef=sway,
                            y0=case["y0"], use_acc_grl=False, **case["kw"])[0]

    def step(steps=nfe):
        out = sample(steps)
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
    assert all(np.isfinite(hw.numpy()).all() for hw in host_wav)

    # ---- correctness of what was just timed (rank 0 holds the fixture's utterance): reference output on the same inputs
    mse = None
    if rank == 0 and fx is not None and a.depth == 22 and a.steps + a.warmup > 0 and not a.outlier_weights:
        if w["golden_steps"] == nfe:
            mse = gen_mse(last["out"].detach().cpu().numpy(), fx["out"], fx)      # the timed region's own last result
        else:                                                        # the fixture is a short solve: one extra untimed sample
            mse = gen_mse(sample(w["golden_steps"]).detach().cpu().numpy(), fx["out"], fx)
        tol = MEL_MSE_TOL
        assert mse <= tol, f"bench.py: the timed path is WRONG: mel-MSE vs the reference's output {mse:.3e} > {tol:g}"

    result = None
    if rank == 0:
        value = world * a.steps * audio_per_step / elapsed
        real_rows, rows_pitch = sum(case["dur_list"]), B * ((N_TOT + 127) // 128 * 128)
        # ragged batches: with --skip-dead 1 the block chain leaves the 128-row blocks that lie wholly in a sample's padding uncomputed (bf16
        # path); the reference -- and the default -- run every sample at the batch's pitch
        ragged = case["lens"] is not None and not a.fp8
        live128 = [(n + 127) // 128 * 128 for n in case["dur_list"]]
        live_ff = [min(v + 128, rows_pitch // B) for v in live128] if a.skip_dead == 2 else live128
        rows_ff = sum(live_ff) if (ragged and a.skip_dead >= 1) else rows_pitch              # rows the FF half of a block computes
        rows_attn = sum(live128) if (ragged and a.skip_masked != 0) else rows_pitch         # rows its attention half computes (exact skipping)
        rows_computed = rows_ff
        flops_step = 2 * nfe * sum(fwd_flops(1, n) for n in case["dur_list"]) * (a.depth / 22.0)
        path_tflops = flops_step / (elapsed / a.steps) / 1e12
        result = {
            "metric": f"audio-seconds/sec @24kHz (NFE={nfe}, CFG on)", "value": value, "unit": "audio-seconds/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8" if a.fp8 else "bf16", "data": "synthetic",
            "rtf": elapsed / (a.steps * audio_per_step),
            "mel_mse_vs_reference": mse,
            "config": {"workload": f"{w['desc']} (F={F_REF}, N={N_TOT}), NFE={nfe}, CFG={CFG}, sway coef {sway} (capped), "
                                   + (("fp8-e4m3 (MXFP8) GEMM operands, " + ("bf16 attention" if a.attn_f8qk == 0 else "attention: Q K^T on MXFP8 q / k (fp8 MFMA), P V bf16"))
                                      if a.fp8 else "bf16 MFMA operands")
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
                       "depth": a.depth, "weights": f"synthetic N(0,0.02^2), seed {w['wseed']}" + (", outlier residual channels (1 % x30)" if a.outlier_weights else ""),
                       "fp8_gemm_sites_kept_bf16": model.engine.stat("fp8_gemms_kept_bf16") if a.fp8 else None,
                       "real_frames_per_step": real_rows, "rows_computed_per_step": rows_computed,
                       "padded_row_waste": 1.0 - real_rows / rows_computed,       # rows the block chain computes beyond the real frames
                       "rows_computed_attention_half": rows_attn,     # attn_norm, QK / V, attention, out-projection: padding blocks always skipped (exact)
                       "rows_at_batch_pitch": rows_pitch, "padded_row_waste_at_batch_pitch": 1.0 - real_rows / rows_pitch,   # what the reference computes
                       "waveforms_per_step": len(segs), "vocode": w["vocode"],
                       "parity_fixture": w["golden"] if mse is not None else None},
            "path_tflops": path_tflops,
            "path_frac": path_tflops / (MFMA_BF16_PEAK_TFLOPS if not a.fp8 else MFMA_FP8_PEAK_TFLOPS),
            "per_rank_ms": {"min": min(per_rank_ms), "max": max(per_rank_ms), "all": [round(x, 3) for x in per_rank_ms]},
        }
        if bcast is not None:
            result["weight_broadcast"] = bcast
        if affinity is not None:
            result["config"]["cpu_affinity"] = affinity

    # ---- roofline of the dominant kernel: short eager pass with per-launch HIP events (rank 0; for N > 1 the other ranks wait at the last
    # barrier meanwhile -- every rank runs the same launches on its own GPU, so rank 0's kernel figures are the job's)
    if rank == 0:
        eng = model.engine
        eng.set_option("profile", 1)
        sample(4)                                  # four Euler steps of the same batch, eager, every big launch stamped by its dispatch
        prof = eng.profile_read()
        eng.set_option("profile", 0)
        lanes = 2 if a.dual else 1                 # dual: each launch covers one CFG branch (B of the 2B branch-rows)
        # algorithmic work of one launch: the REAL frames of the samples it covers (a ragged batch computes more rows than that: the waste
        # is reported, not credited)
        rows = 2.0 * real_rows / lanes
        nsq = 2.0 * sum(n * n for n in case["dur_list"]) / lanes
        bb = 2 * B // lanes
        mm = {k: v for k, v in prof.items() if k in SYMBOL and v[1] > 0}
        by_sym = {}
        for k, (ms, cnt) in mm.items():
            e = by_sym.setdefault(SYMBOL[k], {"ms": 0.0, "launches": 0, "flops": 0.0, "classes": []})
            e["ms"] += ms; e["launches"] += int(cnt); e["flops"] += class_flops(k, rows, nsq) * cnt; e["classes"].append(k)
        dom = max(by_sym, key=lambda s: by_sym[s]["ms"])      # dominant kernel = the symbol with the largest total time
        d = by_sym[dom]
        avg_us = 1e3 * d["ms"] / d["launches"]
        fl = d["flops"] / d["launches"]
        ach = fl / (avg_us * 1e-6) / 1e12
        total_ms = sum(v[0] for v in prof.values())
        # HBM-side bytes per launch, the rocprofv3 average of the same symbol and the counter-backed bound come from COMMITTED profiler
        # passes over this very command (rocprofv3 cannot run inside bench.py); each is labelled with the file it was read from
        def committed(names, *path):
            for name in names:
                try:
                    j = json.load(open(os.path.join(ROOT, "profiles", name[:3], name)))      # profiles/<round>/<file>
                    v = j
                    for k_ in path:
                        v = v[k_]
                    return v, f"profiles/{name[:3]}/{name} <- {j.get('_source', '?')}"
                except (OSError, KeyError, ValueError, TypeError):
                    continue
            return None, None
        wkey = a.workload + ("_fp8" if a.fp8 and a.workload != "configs4" else "")
        traffic_e, traffic_src = committed(("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"), wkey, dom)
        traffic = traffic_e.get("hbm_bytes_per_launch") if traffic_e else None
        rocprof_e, rocprof_src = committed(("r06_kernel_avgs.json", "r05_kernel_avgs.json", "r04_kernel_avgs.json", "r03_kernel_avgs.json"), wkey, dom)
        rocprof_us = rocprof_e.get("avg_us") if rocprof_e else None
        bounds_all, bounds_src = committed(("r06_kernel_bounds.json", "r05_kernel_bounds.json", "r04_kernel_bounds.json"), wkey)
        is_gemm = dom != "attn_fwd_splitkv_kernel"
        peak = MFMA_FP8_PEAK_TFLOPS if (a.fp8 and is_gemm) else MFMA_BF16_PEAK_TFLOPS   # attention stays bf16
        dom_bound = (bounds_all or {}).get(dom)
        # which roof the kernel is priced against follows from its arithmetic intensity (algorithmic flop per algorithmic byte of the launch)
        # against the machine balance (2.5 PFLOP/s over 8 TB/s = 312 flop/B); what holds it BELOW that roof is the counters' verdict
        dom_bytes = sum(class_bytes(k, rows, nsq / max(rows, 1.0)) * mm[k][1] for k in d["classes"]) / d["launches"]
        # HBM-side bytes a launch cannot avoid: its weights (0.37 GB of bf16 block weights per step do not fit the 256 MiB Infinity Cache and stream
        # from HBM every step); the activations it reads and writes (~0.1 GB working set per step) stay in that cache (SURVEY.md 8d prices the path
        # the same way: FLOPs against the in-loop weight bytes).  Attention has no weights: its operands are activations.
        w_bytes = {"gemm_qkv_fused": 3, "gemm_qk_rope": 2, "gemm_v_t": 1, "gemm_attn_out": 1, "gemm_ff1_gelu": 2, "gemm_ff2": 2, "attention": 0}
        dom_wbytes = sum(w_bytes[k] * 1024 * 1024 * (1 if a.fp8 else 2) * mm[k][1] for k in d["classes"]) / d["launches"]
        balance = peak * 1e12 / 8e12
        intensity_hbm = (fl / dom_wbytes) if dom_wbytes else float("inf")
        intensity_all = fl / dom_bytes
        roof = "mfma" if intensity_hbm > balance else "hbm"
        result["roofline"] = {"bound": roof,
                              "bound_from": (f"{fl / 1e9:.2f} GFLOP per launch over {dom_wbytes / 1e6:.1f} MB of weights that must come from HBM = "
                                             f"{intensity_hbm:.0f} flop/B against the machine balance {balance:.0f} flop/B (activations are Infinity-Cache resident); "
                                             f"over ALL operand bytes of the launch ({dom_bytes / 1e6:.1f} MB, the L2 / fabric view) {intensity_all:.0f} flop/B"),
                              "algorithmic_bytes_per_launch": dom_bytes, "hbm_weight_bytes_per_launch": dom_wbytes,
                              "fabric_view": {"achieved": dom_bytes / (avg_us * 1e-6) / 1e9, "unit": "GB/s", "note": "all operand bytes of the launch over its "
                                              "duration: what the L2s / the fabric deliver, not an HBM figure"},
                              "limited_by_verdict": (dom_bound or {}).get("verdict"),
                              "kernel": dom, "classes": d["classes"], "achieved": ach, "peak": peak,
                              "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                              "avg_launch_us": avg_us, "avg_launch_us_source": "this run: eager pass, per-launch HIP event pairs, lanes serialised",
                              "avg_launch_us_rocprof": rocprof_us, "avg_launch_us_rocprof_source": rocprof_src,
                              "frac_rocprof": (fl / (rocprof_us * 1e-6) / 1e12 / peak) if rocprof_us else None,
                              "launches": d["launches"], "flops_per_launch": fl,
                              "time_share": d["ms"] / total_ms,
                              "launch_shape": f"{bb} sample(s), {rows:.0f} real frames per launch ({lanes} concurrent lane(s) in the timed region)",
                              # the MFMA peak is the roof the figure of merit is priced against (arithmetic intensity ~2 700
                              # flop/B); what the kernel is held up BY, from the counters, is `limited_by`
                              "limited_by": dom_bound, "limited_by_source": bounds_src}
        pipes, pipes_src = committed(("r05_kloop_pipes.json", "r04g_kloop_pipes.json"))
        if pipes and dom == "gemm_bf16_kernel<EPI_GATE_RES>" and not a.fp8 and a.workload in ("configs1", "short"):
            # the 128 x 128 tile's K loop taken apart with compile-time ablation builds (measurement only): three pipes that each need about
            # the same time per K-tile, and a loop that overlaps them to 56 %.  Read from the committed JSON that tools/kloop_pipes_json.py
            # derives from the raw profile lines -- no number is typed in here.
            t17 = pipes["tiles"]["17"]
            result["roofline"]["k_loop_pipes"] = {"tile": "128 x 128 x 64, 8 waves, one workgroup per CU (round 4's 8-wave tile; since round 6 this launch runs its "
                                                          "4 compute + 4 loader wave form: same 0.43-0.44 us per K-tile with the compute waves free of LDS-DMA "
                                                          "issue and with a 4-stage ring -- profiles/r06/r06v_kbench_loader_wave_tiles.txt -- i.e. the loop sits on the "
                                                          "LDS port: 32 KB of DMA writes + 64-96 KB of fragment reads per K-tile; 0.7 us less outside the loop)",
                                                  "us_per_k_tile": dict(t17["us_per_k_tile"], **{k + "_ideal": v for k, v in pipes["ideal_us_per_k_tile"].items() if k != "clock_ghz"}),
                                                  "outside_the_loop_us": t17["outside_the_loop_us"], "source": pipes_src}
        if bounds_all:
            result["roofline_kernels"] = {k: v for k, v in bounds_all.items() if not k.startswith("_")}
        # energy: the package sits on its 1.4 kW cap in every workload, so step time follows joules per utterance.  tools/energy_table.py
        # loops each block kernel alone (two lanes) under a rocm-smi sampler; its committed table is read here, nothing is typed in.
        energy_e, energy_src = committed(("r06_energy.json",), a.workload)
        if energy_e and not a.fp8:
            cls = energy_e["classes"]

            def cflops(r):
                return 4.0 * r["M"] * r["M"] * 64 * r["N"] if r["kernel"] == "attention" else 2.0 * r["M"] * r["N"] * r["K"]
            tot_fl = sum(cflops(r) * r["launches_per_step_batch"] for r in cls)
            tot_j = energy_e["sum_j_block_kernels"]
            result["roofline"]["energy"] = {
                "j_per_utterance_measured": energy_e.get("workload", {}).get("j_per_utterance"),
                "package_w": energy_e.get("workload", {}).get("package_w"), "sclk_mhz": energy_e.get("workload", {}).get("sclk_mhz"),
                "table_over_measured": energy_e.get("check", {}).get("ratio"),
                "pj_per_flop_block_kernels": tot_j / tot_fl * 1e12,
                "power_capped_tflops": 1400.0 / (tot_j / tot_fl) / 1e12,      # what 1.4 kW buys at this path's energy per flop
                "by_kernel": {r["class"]: {"j_per_launch": r["j_per_launch"], "package_w": r["package_w"], "sclk_mhz": r["sclk_mhz"],
                                           "us_per_launch_alone_two_lanes": r["us_per_launch"], "pj_per_flop": (r["j_per_launch"] / cflops(r) * 1e12) if cflops(r) else None,
                                           "energy_share": r["share_of_table_j"],
                                           "flop_share": cflops(r) * r["launches_per_step_batch"] / tot_fl} for r in cls},
                "note": "each block kernel looped alone in two concurrent lanes for 3 s under a rocm-smi sampler; J per launch = W x us / lanes; "
                        "the sum over a step batch is checked against W x ms of the workload itself on the same lease (table_over_measured)",
                "source": energy_src}
        g_ms = sum(v["ms"] for s, v in by_sym.items() if s != "attn_fwd_splitkv_kernel")
        g_fl = sum(v["flops"] for s, v in by_sym.items() if s != "attn_fwd_splitkv_kernel")
        gpeak = MFMA_FP8_PEAK_TFLOPS if a.fp8 else MFMA_BF16_PEAK_TFLOPS
        result["roofline_gemm_family"] = {"achieved": g_fl / (g_ms * 1e-3) / 1e12, "peak": gpeak, "frac": g_fl / (g_ms * 1e-3) / 1e12 / gpeak,
                                          "time_share": g_ms / total_ms, "unit": "TFLOP/s"}
        result["kernel_tflops"] = {k: round(class_flops(k, rows, nsq) / (1e3 * v[0] / v[1] * 1e-6) / 1e12, 1) for k, v in mm.items()}
        result["kernel_time_share"] = {k: round(v[0] / total_ms, 4) for k, v in prof.items() if v[1] > 0}
        result["kernel_avg_us"] = {k: round(1e3 * v[0] / v[1], 2) for k, v in prof.items() if v[1] > 0}

        # batched shapes: the in-situ timeline of one ODE step (measurement build, tools/timeline_step.py --json), committed -- per block the sum of
        # the launches' in-situ spans, the gaps between consecutive launches of a lane by site, how much of the step has one / two launches in
        # flight, and which launch is the one that is alone
        tl_e, tl_src = committed((f"r06_timeline_{a.workload}.json",))
        if tl_e and B > 1 and a.dual and not a.fp8:
            result["roofline"]["latency_batched"] = dict({k: v for k, v in tl_e.items() if not k.startswith("_")}, source=tl_src,
                                                         block_us_this_run=None)

        def add_latency_roofline(step_loop_ms, clock_ghz):
            if "latency_batched" in result["roofline"] and a.depth > 0:
                result["roofline"]["latency_batched"]["block_us_this_run"] = 1e3 * step_loop_ms / nfe / a.depth
            # batch 1, two lanes, bf16: the block is a dependent chain per lane, priced against its latency floor next to the MFMA figure
            if pipes and B == 1 and a.dual and not a.fp8 and a.depth > 0:
                result["latency_roofline"] = latency_roofline(result["kernel_avg_us"], pipes, int(rows), 1e3 * step_loop_ms / nfe / a.depth, lanes, clock_ghz)
                result["roofline"]["latency"] = result["latency_roofline"]      # (also inside `roofline`: the driver's record keeps that object whole)

        if a.no_phases or world > 1:
            add_latency_roofline(1e3 * elapsed / a.steps, 2.13)       # (the whole utterance stands in for the step loop: ~3 % more than it)
            if dist:
                dist.barrier()
                dist.destroy_process_group()
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(result) + "\n").encode())
            os.close(json_fd)
            return

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

        ec = model.last_engine_call          # the engine-level inputs of the timed region's CFM.sample call
        prep = lambda: eng.prepare(ec["cond"], ec["cond_mask"], ec["text"], ec["t_grid"], cond_frames=ec["cond_frames"], cfg_strength=ec["cfg_strength"],
                                   seq_len=ec["seq_len"], prosody=ec["prosody"])
        sample(nfe)                                # last_engine_call of a full-NFE solve
        ec = model.last_engine_call
        hoist_ms, _ = timed_ms(prep)
        loop_ms, (o_, _y) = timed_ms(lambda: eng.solve(ec["y0"]), reps=2)
        voc_ms, wav_ = timed_ms(lambda: vocode(o_))
        b0, s0, e0 = segs[0]
        one_mel = o_[b0:b0 + 1, s0:e0, :].permute(0, 2, 1) if w["vocode"] != "generated" else o_[:, s0:e0, :].permute(0, 2, 1)
        dec_ms, wav1 = timed_ms(lambda: vocoder.decode(one_mel))
        d2h_ms, _ = timed_ms(lambda: host_wav[0].copy_(wav1, non_blocking=True))
        voc_ms = max(voc_ms - d2h_ms * len(host_wav), dec_ms)        # vocode() also issued the D2H copies: the decode share of it
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
        L1 = e0 - s0                                                          # frames of the decode timed alone above
        nb1 = one_mel.shape[0]
        voc_bytes = nb1 * (400.0 * L1 + 1024.0 * (L1 - 1)) + wbytes           # weights + mel in + wav out
        voc_traffic, voc_traffic_src = committed(("r06_traffic.json", "r05_traffic.json", "r04_traffic.json"), a.workload, "vocoder")
        voc_hbm = voc_traffic.get("hbm_bytes_per_decode") if voc_traffic else None
        # the decode is fp32 GEMM work (exact fp32 MFMA, 157.3 TFLOP/s dense: MI355X_MICROARCH.md "Peak FP32 (matrix)"), not a byte stream:
        # every weight matrix is applied once per frame, plus the windowed inverse rDFT as a GEMM against its (n_fft + 2) x n_fft basis
        n_fft = int(np.shape(vsd_host["head.istft.window"])[0])
        voc_flops = 2.0 * nb1 * L1 * (sum(int(np.prod(np.shape(v))) for v in vsd_host.values() if np.ndim(v) >= 2) + (n_fft + 2) * n_fft)
        result["roofline_vocoder"] = {"bound": "mfma_f32", "achieved": voc_flops / (dec_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                      "frac": voc_flops / (dec_ms * 1e-3) / 157.3e12, "algorithmic_flops": voc_flops,
                                      "hbm_view": {"achieved": voc_bytes / (dec_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                   "frac": voc_bytes / (dec_ms * 1e-3) / 8e12,
                                                   "note": "SURVEY.md 8d priced this phase against HBM; its 55 MB per decode would take 7 us at 8 TB/s, "
                                                           "the 25 GFLOP of exact-fp32 MFMA work 0.16 ms: the phase is bound by the fp32 matrix rate"},
                                      "algorithmic_bytes": voc_bytes, "decode_ms": dec_ms,
                                      "frames": L1, "batch": nb1, "decodes_per_step": len(segs) if w["vocode"] != "generated" else 1,
                                      "traffic": voc_hbm, "traffic_source": voc_traffic_src,
                                      # both roofs side by side: SURVEY.md 8d's byte pricing and the fp32 matrix rate that actually bounds exact-fp32 GEMMs
                                      "frac_vs_hbm_roof": voc_bytes / (dec_ms * 1e-3) / 8e12, "frac_vs_fp32_mfma_roof": voc_flops / (dec_ms * 1e-3) / 157.3e12,
                                      "traffic_over_algorithmic": (voc_hbm / voc_bytes) if voc_hbm else None,
                                      "traffic_note": "the launches round-trip the [L, 512] / [L, 1536] fp32 activations between them: HBM-side traffic is this many "
                                                      "times the algorithmic bytes (weights + mel in + waveform out)",
                                      "note": "a chain of 36 small exact-fp32 launches replayed as one hipGraph; the GEMMs run at 40-60 % of the fp32 MFMA rate, the rest is launch boundaries and the dwconv / LN / overlap-add launches"}
        v_ = result["roofline_vocoder"]
        result["roofline"]["vocoder"] = {k: v_[k] for k in ("frac_vs_hbm_roof", "frac_vs_fp32_mfma_roof", "traffic", "traffic_over_algorithmic", "algorithmic_bytes",
                                                            "algorithmic_flops", "decode_ms", "frames", "traffic_source")}
        result["clock_power"] = None if a.no_clock_power else clock_power(step)
        if result["clock_power"]:
            # the MFMA peak the roofline prices against is the 2.4 GHz figure; what the package was clocked to deliver while this ran
            result["roofline"]["peak_at_measured_clock"] = result["roofline"]["peak"] * result["clock_power"]["sclk_mhz"] / 2400.0
        add_latency_roofline(loop_ms, (result["clock_power"] or {}).get("sclk_mhz", 2130.0) / 1e3)
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
