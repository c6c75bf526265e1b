#!/usr/bin/env python
"""First contact with REAL artefacts: every parity number of this repository is on synthetic N(0, 0.02^2) weights, because
no checkpoint ships with the reference tree.  The day the real files are at hand, this prints a pass / fail checklist of everything that could
not be pinned in the build image:

    python tools/first_contact.py --model multilingual_grl --ckpt <ckpt.safetensors|.pt> --vocab <vocab.txt> --vocos <vocos dir>
                                  [--pretssel-cfg <json> --prosody-ckpt <pt>] [--uvr5 <dir with Kim_Vocal_1.onnx + json | network file>]
                                  [--prompt <wav>] [--no-ema] [--device cuda:0]

  1 checkpoint      strict load of the CFM state dict against the layout this build declares (the reference's key surgery first:
                    lemas_tts/infer/utils_infer.py:215-241): missing / unexpected / mis-shaped tensors by name
  2 rope            the RoPE convention probe.  x_transformers' rotary embedding (interleaved pairs (x0,x1)(x2,x3)...) is third
                    party and could not be pinned; the alternative is the half-split form (x_i, x_{i+32}).  With TRAINED weights the right
                    convention gives the first blocks' attention its locality (mass near the diagonal) and a lower entropy; the wrong one
                    scrambles relative position.  Runs the first two DiT blocks' q / k on the CPU in fp32 (torch, this file's own few lines --
                    neither the product nor the test oracle) under both conventions on a prompt mel (--prompt, else a synthetic one).
                    On synthetic weights both conventions look alike: INCONCLUSIVE, not a failure.
  3 vocos           config.yaml against VocosArch, pytorch_model.bin keys / shapes against the layout (feature_extractor.* ignored), the stored
                    ISTFT window against hann(n_fft)
  4 prosody         pretssel_cfg.json "model.prosody_*" keys against ProsodyArch's assumed defaults, checkpoint keys against the layout
  5 uvr5            Kim_Vocal_1.onnx read without onnx / onnxruntime: node histogram, the ConvTDFNet hyper-parameters the graph implies against
                    the published values this build assumes, json configuration against MDXConfig's defaults, parameter count
  6 device          (only with a GPU) engines built from the real files; one 2-step synthesis and one denoiser forward give finite output
Exit code 0 when nothing FAILED (SKIP and INCONCLUSIVE are not failures)."""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RESULTS = []


def report(item, status, detail=""):
    RESULTS.append((item, status))
    print(f"[{status:^12s}] {item}" + (f": {detail}" if detail else ""), flush=True)


def compare_layout(item, have: dict, want: dict, ignore=()):
    have = {k: tuple(v.shape) for k, v in have.items() if not k.startswith(tuple(ignore))}
    missing = [k for k in want if k not in have]
    unexpected = [k for k in have if k not in want]
    shapes = [f"{k} {have[k]} != {tuple(want[k])}" for k in want if k in have and have[k] != tuple(want[k])]
    if missing or unexpected or shapes:
        report(item, "FAIL", f"{len(missing)} missing {missing[:5]}, {len(unexpected)} unexpected {unexpected[:5]}, {len(shapes)} mis-shaped {shapes[:3]}")
        return False
    report(item, "PASS", f"{len(want)} tensors, {sum(int(np.prod(s)) for s in want.values()) / 1e6:.1f} M values")
    return True


# ---- 2: RoPE probe (fp32 torch on the CPU; the DiT arithmetic of dit.py:93-99, modules.py:167-190, 310-315, 452-480 in a few lines) ----------
def _rope(t, conv):
    n, d = t.shape[-2], t.shape[-1]
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2).float() / d))
    ang = torch.outer(torch.arange(n).float(), inv)                      # [n, d/2]
    if conv == "interleaved":                                            # pairs (x0,x1),(x2,x3)...: x_transformers >= 1.31 / GPT-J
        cos, sin = ang.cos().repeat_interleave(2, -1), ang.sin().repeat_interleave(2, -1)
        rot = torch.stack((-t[..., 1::2], t[..., 0::2]), dim=-1).reshape(t.shape)
    else:                                                                # pairs (x_i, x_{i+d/2}): GPT-NeoX / half-split
        cos, sin = torch.cat((ang.cos(), ang.cos()), -1), torch.cat((ang.sin(), ang.sin()), -1)
        rot = torch.cat((-t[..., d // 2:], t[..., :d // 2]), dim=-1)
    return t * cos + rot * sin


def rope_probe(sd, arch, mel, blocks=2):
    import torch.nn.functional as F
    g = lambda k: sd["transformer." + k].float()
    n = mel.shape[0]
    d, H, dh = arch.dim, arch.heads, arch.dim_head
    x0 = torch.randn(n, arch.mel_dim, generator=torch.Generator().manual_seed(0))
    text = torch.zeros(n, arch.text_dim)
    x = F.linear(torch.cat((x0, mel, text), -1), g("input_embed.proj.weight"), g("input_embed.proj.bias"))
    c = x.t()[None]
    for i in (0, 2):
        c = F.mish(F.conv1d(c, g(f"input_embed.conv_pos_embed.conv1d.{i}.weight"), g(f"input_embed.conv_pos_embed.conv1d.{i}.bias"),
                            padding=arch.conv_pos_kernel // 2, groups=arch.conv_pos_groups))
    x = x + c[0].t()
    half = arch.time_freq_dim // 2
    fr = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
    temb = torch.cat(((1000 * 0.5 * fr).sin(), (1000 * 0.5 * fr).cos()))  # t = 0.5
    temb = F.linear(F.silu(F.linear(temb, g("time_embed.time_mlp.0.weight"), g("time_embed.time_mlp.0.bias"))),
                    g("time_embed.time_mlp.2.weight"), g("time_embed.time_mlp.2.bias"))
    out = {}
    for conv in ("interleaved", "half_split"):
        h, ent, loc = x.clone(), [], []
        for b in range(blocks):
            p = f"transformer_blocks.{b}."
            mod = F.linear(F.silu(temb), g(p + "attn_norm.linear.weight"), g(p + "attn_norm.linear.bias")).chunk(6)
            hn = F.layer_norm(h, (d,), eps=1e-6) * (1 + mod[1]) + mod[0]
            q = F.linear(hn, g(p + "attn.to_q.weight"), g(p + "attn.to_q.bias")).view(n, H, dh).transpose(0, 1)
            k = F.linear(hn, g(p + "attn.to_k.weight"), g(p + "attn.to_k.bias")).view(n, H, dh).transpose(0, 1)
            v = F.linear(hn, g(p + "attn.to_v.weight"), g(p + "attn.to_v.bias")).view(n, H, dh).transpose(0, 1)
            a = torch.softmax(_rope(q, conv) @ _rope(k, conv).transpose(-1, -2) / math.sqrt(dh), -1)     # [H, n, n]
            ent.append(float(-(a * (a + 1e-30).log()).sum(-1).mean()))
            idx = torch.arange(n)
            band = ((idx[:, None] - idx[None, :]).abs() <= 32).float()
            loc.append(float((a * band).sum(-1).mean()))
            o = (a @ v).transpose(0, 1).reshape(n, H * dh)
            h = h + mod[2] * F.linear(o, g(p + "attn.to_out.0.weight"), g(p + "attn.to_out.0.bias"))
            hn = F.layer_norm(h, (d,), eps=1e-6) * (1 + mod[4]) + mod[3]
            ff = F.linear(F.gelu(F.linear(hn, g(p + "ff.ff.0.0.weight"), g(p + "ff.ff.0.0.bias")), approximate="tanh"),
                          g(p + "ff.ff.2.weight"), g(p + "ff.ff.2.bias"))
            h = h + mod[5] * ff
        out[conv] = {"entropy_nats": sum(ent) / len(ent), "mass_within_32_frames": sum(loc) / len(loc)}
    out["uniform_entropy_nats"] = math.log(n)
    out["uniform_mass_within_32_frames"] = float(min(65.0 / n, 1.0))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model", default="multilingual_grl")
    ap.add_argument("--ckpt")
    ap.add_argument("--vocab")
    ap.add_argument("--vocos")
    ap.add_argument("--pretssel-cfg")
    ap.add_argument("--prosody-ckpt")
    ap.add_argument("--uvr5")
    ap.add_argument("--prompt")
    ap.add_argument("--no-ema", action="store_true")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--depth", type=int, default=None, help="override the yaml's depth (synthetic test files)")
    a = ap.parse_args(argv)
    RESULTS.clear()
    from lemas_tts_amd.infer import utils_infer as U
    from lemas_tts_amd.model.layout import DiTArch, ProsodyArch, VocosArch, cfm_param_shapes, prosody_param_shapes, vocos_param_shapes
    gpu = torch.cuda.is_available()
    cfg = U.load_arch_config(a.model)
    arch_d = dict(cfg["arch"])
    if a.depth is not None:
        arch_d["depth"] = a.depth
    arch = DiTArch.from_yaml_arch(arch_d)
    sd = None

    # ---- 1 checkpoint
    if a.ckpt:
        try:
            sd = U.read_checkpoint(a.ckpt, use_ema=not a.no_ema)
            vocab_size = None
            if a.vocab:
                _, vocab_size = U.get_tokenizer(a.vocab, "custom")
            rows = sd["transformer.text_embed.text_embed.weight"].shape[0] if "transformer.text_embed.text_embed.weight" in sd else None
            if vocab_size is None and rows:
                vocab_size = rows - 1
                report("1 vocab", "SKIP", f"no --vocab: {vocab_size} tokens taken from the embedding table's {rows} rows")
            elif rows is not None:
                report("1 vocab", "PASS" if rows == vocab_size + 1 else "FAIL", f"vocab.txt {vocab_size} entries, embedding table {rows} rows (want entries + 1 filler)")
            prosody = any(k.startswith(("prosody_to_mel.", "transformer.prosody_text_proj.")) for k in sd)
            ok = compare_layout("1 checkpoint strict load" + (" (prosody model)" if prosody else ""), sd, cfm_param_shapes(arch, vocab_size, prosody),
                                ignore=("prosody_encoder.",))
            inv = sd.get("transformer.rotary_embed.inv_freq")
            if inv is not None:
                want = 1.0 / (10000.0 ** (torch.arange(0, arch.dim_head, 2).float() / arch.dim_head))
                report("1 rotary inv_freq buffer", "PASS" if torch.allclose(inv.float(), want, rtol=1e-5) else "FAIL", f"theta 10000, {inv.numel()} frequencies")
            if not ok:
                sd = None
        except Exception as e:   # noqa: BLE001
            report("1 checkpoint strict load", "FAIL", f"{type(e).__name__}: {e}")
            sd = None
    else:
        report("1 checkpoint strict load", "SKIP", "no --ckpt")

    # ---- 2 rope
    if sd is not None:
        try:
            if a.prompt:
                from lemas_tts_amd.infer.audio_io import load_wav
                wav, sr = load_wav(a.prompt)
                if gpu:
                    from lemas_tts_amd.engine import MelEngine, resampler
                    w = wav.mean(0, keepdim=True).to(a.device)
                    w = resampler(sr, 24000, device=a.device)(w) if sr != 24000 else w
                    mel = MelEngine(device=a.device).frames_first(w)[0].cpu()[:400]
                else:
                    raise RuntimeError("--prompt needs the GPU mel front end")
            else:
                from lemas_tts_amd import synth
                mel = torch.from_numpy(synth.synth_cond_mel(3, 300))
            r = rope_probe({k: v for k, v in sd.items()}, arch, mel.float())
            il, hs = r["interleaved"], r["half_split"]
            detail = (f"interleaved: entropy {il['entropy_nats']:.3f} nats, local mass {il['mass_within_32_frames']:.3f}; half-split: "
                      f"{hs['entropy_nats']:.3f}, {hs['mass_within_32_frames']:.3f}; uniform attention: {r['uniform_entropy_nats']:.3f}, "
                      f"{r['uniform_mass_within_32_frames']:.3f}")
            better_loc = il["mass_within_32_frames"] / max(hs["mass_within_32_frames"], 1e-9)
            if better_loc > 1.15 and il["entropy_nats"] < hs["entropy_nats"]:
                report("2 rope convention (interleaved pairs, what this build implements)", "PASS", detail)
            elif better_loc < 1 / 1.15 and hs["entropy_nats"] < il["entropy_nats"]:
                report("2 rope convention", "FAIL", "the HALF-SPLIT convention is the one that gives attention its locality -- csrc/gemm_bf16.hip EPI_QK_ROPE "
                       "and oracle/ref_shims.py rotate the wrong pairs; " + detail)
            else:
                report("2 rope convention", "INCONCLUSIVE", "the two conventions are indistinguishable on these weights (expected for synthetic ones); " + detail)
        except Exception as e:   # noqa: BLE001
            report("2 rope convention", "FAIL", f"{type(e).__name__}: {e}")
    else:
        report("2 rope convention", "SKIP", "needs a checkpoint that loads")

    # ---- 3 vocos
    vsd, varch = None, None
    if a.vocos:
        try:
            import yaml
            with open(os.path.join(a.vocos, "config.yaml")) as f:
                vc = yaml.safe_load(f)
            bb, hd = vc["backbone"]["init_args"], vc["head"]["init_args"]
            varch = VocosArch(input_channels=bb["input_channels"], dim=bb["dim"], intermediate_dim=bb["intermediate_dim"], num_layers=bb["num_layers"],
                              n_fft=hd["n_fft"], hop_length=hd["hop_length"])
            report("3 vocos config.yaml", "PASS" if varch == VocosArch() else "INCONCLUSIVE",
                   f"{varch}" + ("" if varch == VocosArch() else f" differs from the assumed {VocosArch()}: the engine is built from the file's values"))
            if hd.get("padding", "center") != "center":
                report("3 vocos ISTFT padding", "FAIL", f"padding '{hd.get('padding')}': only 'center' is built (SURVEY.md 8a-V)")
            vsd = torch.load(os.path.join(a.vocos, "pytorch_model.bin"), map_location="cpu", weights_only=True)
            compare_layout("3 vocos strict load", vsd, vocos_param_shapes(varch), ignore=("feature_extractor.",))
            win = vsd.get("head.istft.window")
            if win is not None:
                report("3 vocos ISTFT window", "PASS" if torch.allclose(win.float(), torch.hann_window(varch.n_fft), atol=1e-6) else "FAIL", "hann(n_fft), periodic")
        except Exception as e:   # noqa: BLE001
            report("3 vocos", "FAIL", f"{type(e).__name__}: {e}")
            vsd = None
    else:
        report("3 vocos", "SKIP", "no --vocos")

    # ---- 4 prosody
    if a.pretssel_cfg:
        try:
            pc = json.load(open(a.pretssel_cfg))
            parch = ProsodyArch.from_pretssel_cfg(pc["model"])
            report("4 pretssel_cfg.json", "PASS" if parch == ProsodyArch() else "INCONCLUSIVE",
                   "equals the published Pretssel values this build assumed" if parch == ProsodyArch() else f"{parch} differs from the assumed defaults (the engine reads the file)")
            if a.prosody_ckpt:
                from lemas_tts_amd.model.prosody_encoder import _strip_state
                psd = _strip_state(dict(torch.load(a.prosody_ckpt, map_location="cpu", weights_only=True)))
                compare_layout("4 prosody encoder strict load", psd, prosody_param_shapes(parch))
        except Exception as e:   # noqa: BLE001
            report("4 prosody", "FAIL", f"{type(e).__name__}: {e}")
    else:
        report("4 prosody", "SKIP", "no --pretssel-cfg")

    # ---- 5 uvr5
    mdx = None
    if a.uvr5:
        try:
            from lemas_tts_amd.uvr5 import MDXConfig, mdx as M, onnx_weights as OW
            from lemas_tts_amd.uvr5.arch import KIM_VOCAL_1, MdxArch, flops
            if os.path.isdir(a.uvr5):
                path, mcfg = M.resolve_model_dir(a.uvr5)
            else:
                path, mcfg = a.uvr5, MDXConfig()
            report("5 uvr5 configuration", "PASS" if (mcfg.mdx_dim_f_set, mcfg.mdx_dim_t_set, mcfg.mdx_n_fft_scale_set) == (3072, 8, 7680) else "INCONCLUSIVE",
                   f"dim_f {mcfg.mdx_dim_f_set}, dim_t 2^{mcfg.mdx_dim_t_set}, n_fft {mcfg.mdx_n_fft_scale_set}, compensate {mcfg.compensate}, is_denoise {mcfg.is_denoise} "
                   "(published Kim_Vocal_1 values assumed: 3072 / 2^8 / 7680)")
            if path.lower().endswith(".onnx"):
                g = OW.read_onnx(path)
                hist = {}
                for n_ in g.nodes:
                    hist[n_.op] = hist.get(n_.op, 0) + 1
                report("5 uvr5 onnx graph", "PASS", f"{len(g.nodes)} nodes {dict(sorted(hist.items()))}, {len(g.tensors)} constants, input {g.inputs}")
            march, msd = OW.load_network_file(path, dim_t=2 ** mcfg.mdx_dim_t_set)
            ma = MdxArch(**march)
            same = ma == KIM_VOCAL_1
            report("5 uvr5 network hyper-parameters", "PASS" if same else "INCONCLUSIVE",
                   f"{ma}; {sum(v.size for v in msd.values()) / 1e6:.2f} M values, {flops(ma) / 1e12:.3f} TFLOP per chunk"
                   + ("" if same else f" -- differs from the assumed {KIM_VOCAL_1} (the engine is built from the file)"))
            if ma.dim_f != mcfg.mdx_dim_f_set:
                report("5 uvr5 network vs configuration", "FAIL", f"the network's dim_f {ma.dim_f} is not the configuration's {mcfg.mdx_dim_f_set}")
            mdx = (ma, msd, mcfg)
        except Exception as e:   # noqa: BLE001
            report("5 uvr5", "FAIL", f"{type(e).__name__}: {e}")
    else:
        report("5 uvr5", "SKIP", "no --uvr5")

    # ---- 6 device
    if not gpu:
        report("6 device", "SKIP", "no GPU in this process")
    else:
        try:
            if sd is not None and vsd is not None:
                from lemas_tts_amd import synth
                from lemas_tts_amd.engine import VocosEngine
                from lemas_tts_amd.model.cfm import CFM
                vs = sd["transformer.text_embed.text_embed.weight"].shape[0] - 1
                prosody = any(k.startswith("prosody_to_mel.") for k in sd)
                m = CFM(arch=arch, vocab_size=vs, state_dict={k: v for k, v in sd.items() if not k.startswith("prosody_encoder.")}, device=a.device,
                        use_prosody_encoder=False) if not prosody else None
                if m is not None:
                    cond = torch.from_numpy(synth.synth_cond_mel(5, 120))[None].to(a.device)
                    text = torch.from_numpy(synth.synth_tokens(6, 40, vs))[None].to(a.device)
                    out, _ = m.sample(cond, text, 240, steps=2, cfg_strength=2.0, sway_sampling_coef=None, seed=0, use_acc_grl=False)
                    wav = VocosEngine(vsd, device=a.device, arch=varch).decode(out[:, 120:].transpose(1, 2).float().contiguous())
                    ok = bool(torch.isfinite(out).all() and torch.isfinite(wav).all())
                    report("6 device: 2-step synthesis on the real weights", "PASS" if ok else "FAIL", f"mel {tuple(out.shape)}, wav {tuple(wav.shape)}, |wav| max {float(wav.abs().max()):.3f}")
                else:
                    report("6 device: synthesis", "SKIP", "prosody model: needs a prosody embedding")
            if mdx is not None:
                from lemas_tts_amd.engine import MdxEngine
                ma, msd, _ = mdx
                e = MdxEngine(ma, msd, device=a.device)
                y = e.forward(torch.randn(1, ma.dim_c, ma.dim_f, ma.dim_t, device=a.device))
                report("6 device: MDX-Net forward on the real weights", "PASS" if bool(torch.isfinite(y).all()) else "FAIL", f"rms {float(y.pow(2).mean().sqrt()):.4f}")
        except Exception as e:   # noqa: BLE001
            report("6 device", "FAIL", f"{type(e).__name__}: {e}")

    failed = [i for i, s in RESULTS if s == "FAIL"]
    print(f"\n{len(RESULTS)} checks: {sum(s == 'PASS' for _, s in RESULTS)} passed, {len(failed)} FAILED, "
          f"{sum(s == 'INCONCLUSIVE' for _, s in RESULTS)} inconclusive, {sum(s == 'SKIP' for _, s in RESULTS)} skipped")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
