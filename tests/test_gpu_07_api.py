"""GPU tier: the mirrored call surface end to end -- TTS.infer (two text lines -> two generations, vocoder, rms
rescale, cross-fade) on synthetic weights against the same pipeline assembled from the oracle."""
import numpy as np
import pytest
import torch

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch

pytestmark = pytest.mark.gpu


def test_tts_infer_end_to_end_vs_oracle():
    from lemas_tts_amd.api import TTS
    from lemas_tts_amd.infer.utils_infer import cross_fade_concat
    from oracle import lemas_oracle as O
    import lemas_tts_amd.infer.utils_infer as UI

    arch = DiTArch(depth=2)
    vocab = {f"p{i}": i for i in range(898)}
    sd = synth.synth_cfm_state_dict(arch, 898, 31)
    vsd = synth.synth_vocos_state_dict(32)
    # depth-2 stand-in for the yaml's depth 22 (weights are synthetic anyway)
    UI_load = UI.load_arch_config
    UI.load_arch_config = lambda m: {**UI_load(m), "arch": {**UI_load(m)["arch"], "depth": 2}}
    import lemas_tts_amd.api as A
    A.load_arch_config = UI.load_arch_config
    try:
        tts = TTS(model="multilingual_grl", device="cuda:0", state_dict=sd, vocoder_state_dict=vsd, vocab_char_map=vocab)
    finally:
        UI.load_arch_config = UI_load
        A.load_arch_config = UI_load
    F_ = 60
    ref_mel = torch.from_numpy(synth.synth_cond_mel(33, F_))
    ref_text = [f"p{i}" for i in synth.synth_tokens(34, 12, 898)]
    lines = [[f"p{i}" for i in synth.synth_tokens(35 + k, 9 + 3 * k, 898)] for k in range(2)]
    ref_len = F_ - 1
    durs = [ref_len + int(ref_len / len(ref_text) * len(g) / 1.0) for g in lines]
    durs = [max(d, F_ + 1, len(ref_text) + len(g) + 1) for d, g in zip(durs, lines)]
    noise = [torch.from_numpy(synth.synth_noise(40 + k, d))[None] for k, d in enumerate(durs)]
    wav, sr, spec = tts.infer(ref_mel, ref_text, lines, nfe_step=4, cfg_strength=2, sway_sampling_coef=5, noise=noise, seed=1)
    assert sr == 24000

    cfm, voc = O.OracleCFM(sd, arch), O.OracleVocos(vsd)
    waves, mels = [], []
    for k, g in enumerate(lines):
        text = O.tokens_to_idx([ref_text + g], vocab)
        dur = ref_len + int(ref_len / len(ref_text) * len(g) / 1.0)
        out, _ = cfm.sample(ref_mel[None], text, dur, y0=noise[k], steps=4, cfg_strength=2, sway_sampling_coef=5)
        mel = out[:, ref_len:, :].permute(0, 2, 1)
        mels.append(mel[0].numpy())
        waves.append(voc.decode(mel)[0].numpy())
    ref_wav = np.clip(cross_fade_concat(waves, 0.15), -0.999, 0.999)
    ref_spec = np.concatenate(mels, axis=1)
    assert wav.shape == ref_wav.shape and spec.shape == ref_spec.shape
    mse = float(((spec - ref_spec) ** 2).mean())
    rel = float(np.sqrt(((wav - ref_wav) ** 2).mean()) / np.sqrt((ref_wav ** 2).mean()))
    print(f"\n[tts.infer] mel-MSE {mse:.3e}  waveform relative rms error {rel:.3e}")
    assert mse <= 1e-4
    assert rel < 5e-2      # bf16-level mel differences pushed through a random-weight vocoder


def test_infer_batch_process_from_raw_audio_with_rms_rescale():
    """utils_infer.py:487-497,552-553: quiet 24 kHz mono/stereo prompt -> boosted to rms 0.1 for the model, output scaled back."""
    import math
    from lemas_tts_amd.infer.utils_infer import infer_batch_process, load_model, load_vocoder
    from oracle import lemas_oracle as O
    arch = DiTArch(depth=1)
    vocab = {f"p{i}": i for i in range(898)}
    sd = synth.synth_cfm_state_dict(arch, 898, 91)
    vsd = synth.synth_vocos_state_dict(92)
    model = load_model(None, dict(dim=1024, depth=1, heads=16, ff_mult=2, text_dim=512, conv_layers=4), "", device="cuda:0",
                       state_dict=sd, vocab_char_map=vocab)
    vocoder = load_vocoder("vocos", device="cuda:0", state_dict=vsd)
    nw = 256 * 50 + 100
    t = torch.arange(nw) / 24000.0
    mono = 0.02 * torch.sin(2 * math.pi * 330.0 * t) + 0.004 * torch.randn(nw, generator=torch.Generator().manual_seed(1))
    stereo = torch.stack([mono * 1.1, mono * 0.9])          # averaged back to `mono` (:488-489)
    ref_text = [f"p{i}" for i in synth.synth_tokens(93, 10, 898)]
    gen = [[f"p{i}" for i in synth.synth_tokens(94, 8, 898)]]
    ref_len = nw // 256
    dur = ref_len + int(ref_len / len(ref_text) * len(gen[0]) / 1.0)
    N = max(dur, ref_len + 2)
    noise = [torch.from_numpy(synth.synth_noise(95, N))[None]]
    wav, sr, spec = next(infer_batch_process((stereo, 24000), ref_text, gen, model, vocoder, nfe_step=3, cfg_strength=2.0,
                                             sway_sampling_coef=5, use_acc_grl=False, noise=noise))
    rms = float(torch.sqrt(torch.mean(mono ** 2)))
    assert rms < 0.1
    boosted = mono * 0.1 / rms
    mel = O.vocos_mel_spectrogram(boosted[None]).permute(0, 2, 1)
    out, _ = O.OracleCFM(sd, arch).sample(mel, O.tokens_to_idx([ref_text + gen[0]], vocab), dur, y0=noise[0], steps=3,
                                          cfg_strength=2.0, sway_sampling_coef=5)
    gmel = out[:, ref_len:, :].permute(0, 2, 1)
    ref_wav = np.clip((O.OracleVocos(vsd).decode(gmel)[0] * rms / 0.1).numpy(), -0.999, 0.999)
    assert wav.shape == ref_wav.shape and sr == 24000
    mse = float(((spec - gmel[0].numpy()) ** 2).mean())
    rel = float(np.sqrt(((wav - ref_wav) ** 2).mean()) / np.sqrt((ref_wav ** 2).mean()))
    print(f"\n[infer_batch_process raw audio] mel-MSE {mse:.3e}  waveform relative rms error {rel:.3e}")
    assert mse <= 1e-4 and rel < 5e-2


def test_infer_batch_process_batched_lines_vs_oracle_batch():
    """three lines of unequal length as ONE CFM.sample batch (``batch_lines=3``) -- against the oracle
    sampling the same batch (the reference's B > 1 semantics: `lens`, duration masks), and equal-length lines against
    the serial path bit for bit (what the data-parallel split relies on)."""
    from lemas_tts_amd.engine import VocosEngine
    from lemas_tts_amd.infer.utils_infer import cross_fade_concat, infer_batch_process
    from lemas_tts_amd.model.cfm import CFM
    from oracle import lemas_oracle as O

    arch = DiTArch(depth=2)
    vocab = {f"p{i}": i for i in range(898)}
    sd = synth.synth_cfm_state_dict(arch, 898, 81)
    vsd = synth.synth_vocos_state_dict(82)
    model = CFM(arch, 898, sd, vocab_char_map=vocab, device="cuda:0")

    class _V:            # what load_vocoder returns: an object with .engine
        engine = VocosEngine(vsd, device="cuda:0")
    F_ = 80
    ref_mel = torch.from_numpy(synth.synth_cond_mel(83, F_))
    ref_text = [f"p{i}" for i in synth.synth_tokens(84, 14, 898)]
    lines = [[f"p{i}" for i in synth.synth_tokens(85 + k, n, 898)] for k, n in enumerate((9, 17, 12))]
    ref_len = F_ - 1
    durs = [ref_len + int(ref_len / len(ref_text) * len(g)) for g in lines]
    noise = [torch.from_numpy(synth.synth_noise(90 + k, d))[None] for k, d in enumerate(durs)]
    kw = dict(nfe_step=3, cfg_strength=2.0, sway_sampling_coef=5, use_acc_grl=False)
    wav, sr, spec = next(infer_batch_process(ref_mel, ref_text, lines, model, _V, noise=noise, batch_lines=3, **kw))

    text = O.tokens_to_idx([ref_text + g for g in lines], vocab)
    y0 = torch.nn.utils.rnn.pad_sequence([n[0] for n in noise], batch_first=True)
    out, _ = O.OracleCFM(sd, arch).sample(ref_mel[None].expand(3, -1, -1), text, torch.tensor(durs), y0=y0, steps=3, cfg_strength=2.0,
                                          sway_sampling_coef=5)
    voc = O.OracleVocos(vsd)
    mels = [out[j: j + 1, ref_len: durs[j], :].permute(0, 2, 1) for j in range(3)]
    ref_spec = np.concatenate([m[0].numpy() for m in mels], axis=1)
    ref_wav = np.clip(cross_fade_concat([voc.decode(m)[0].numpy() for m in mels], 0.15), -0.999, 0.999)
    assert spec.shape == ref_spec.shape and wav.shape == ref_wav.shape
    mse = float(((spec - ref_spec) ** 2).mean())
    rel = float(np.sqrt(((wav - ref_wav) ** 2).mean()) / np.sqrt((ref_wav ** 2).mean()))
    print(f"\n[batch_lines=3 vs oracle batch] mel-MSE {mse:.3e}  waveform relative rms error {rel:.3e}")
    assert mse <= 1e-4 and rel < 5e-2

    # lines of very unequal length (durations 129 / 400 / 259 frames: 2, 4 and 3 live 128-row blocks of a 512-row pitch) with the padding
    # blocks left uncomputed (skip_padding_blocks -> engine option skip_dead): still inside the tolerance against the oracle's batch
    lines2 = [[f"p{i}" for i in synth.synth_tokens(185 + k, n, 898)] for k, n in enumerate((9, 57, 32))]
    durs2 = [ref_len + int(ref_len / len(ref_text) * len(g)) for g in lines2]
    noise2 = [torch.from_numpy(synth.synth_noise(190 + k, d))[None] for k, d in enumerate(durs2)]
    _, _, spec2 = next(infer_batch_process(ref_mel, ref_text, lines2, model, _V, noise=noise2, batch_lines=3, skip_padding_blocks=True, **kw))
    text2 = O.tokens_to_idx([ref_text + g for g in lines2], vocab)
    y02 = torch.nn.utils.rnn.pad_sequence([n[0] for n in noise2], batch_first=True)
    out2, _ = O.OracleCFM(sd, arch).sample(ref_mel[None].expand(3, -1, -1), text2, torch.tensor(durs2), y0=y02, steps=3, cfg_strength=2.0,
                                           sway_sampling_coef=5)
    ref_spec2 = np.concatenate([out2[j, ref_len: durs2[j], :].T.numpy() for j in range(3)], axis=1)
    assert spec2.shape == ref_spec2.shape
    mse2 = float(((spec2 - ref_spec2) ** 2).mean())
    print(f"[batch_lines=3, padding blocks skipped, durations {durs2}] mel-MSE {mse2:.3e}")
    assert mse2 <= 1e-4
    model.engine.set_option("skip_dead", 0)

    # equal lengths: batched == serial, bit for bit
    same = [lines[1], [f"p{i}" for i in synth.synth_tokens(99, len(lines[1]), 898)]]
    nz = [noise[1], torch.from_numpy(synth.synth_noise(98, durs[1]))[None]]
    a = next(infer_batch_process(ref_mel, ref_text, same, model, _V, noise=nz, batch_lines=2, **kw))
    b = next(infer_batch_process(ref_mel, ref_text, same, model, _V, noise=nz, batch_lines=1, **kw))
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[0], b[0])


def test_tts_infer_seed_reproduces_the_reference_noise_sequence():
    """api.py:194-197 seeds python / torch once; every generated line then draws its y0 from that host generator in order
    (utils_infer.py:531-542 passes no seed to the sampler).  TTS.infer(seed=S) must equal the pipeline fed with exactly those
    draws, and two calls with the same seed must agree bit for bit."""
    from lemas_tts_amd.api import TTS
    import lemas_tts_amd.api as A
    import lemas_tts_amd.infer.utils_infer as UI
    arch_cfg = UI.load_arch_config
    A.load_arch_config = lambda m: {**arch_cfg(m), "arch": {**arch_cfg(m)["arch"], "depth": 2}}
    try:
        vocab = {f"p{i}": i for i in range(898)}
        tts = TTS(model="multilingual_grl", device="cuda:0", state_dict=synth.synth_cfm_state_dict(DiTArch(depth=2), 898, 61),
                  vocoder_state_dict=synth.synth_vocos_state_dict(62), vocab_char_map=vocab)
    finally:
        A.load_arch_config = arch_cfg
    F_ = 50
    ref_mel = torch.from_numpy(synth.synth_cond_mel(63, F_))
    ref_text = [f"p{i}" for i in synth.synth_tokens(64, 10, 898)]
    lines = [[f"p{i}" for i in synth.synth_tokens(65 + k, 8 + 2 * k, 898)] for k in range(2)]
    kw = dict(nfe_step=2, cfg_strength=2, sway_sampling_coef=5)
    a = tts.infer(ref_mel, ref_text, lines, seed=77, **kw)
    b = tts.infer(ref_mel, ref_text, lines, seed=77, **kw)
    np.testing.assert_array_equal(a[0], b[0])
    ref_len = F_ - 1
    durs = [max(ref_len + int(ref_len / len(ref_text) * len(g)), F_ + 1, len(ref_text) + len(g) + 1) for g in lines]
    torch.manual_seed(77)
    noise = [torch.randn(d, 100)[None] for d in durs]               # the draws the reference would make, in order
    c = tts.infer(ref_mel, ref_text, lines, seed=5, noise=noise, **kw)
    np.testing.assert_array_equal(a[2], c[2])
    np.testing.assert_array_equal(a[0], c[0])
