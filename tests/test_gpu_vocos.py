"""GPU tier: Vocos decode through the C ABI against the oracle restatement (which itself is checked against
torch.istft on CPU).  Everything is fp32 on the GPU (f32 MFMA), so the waveform tolerance is tight."""
import numpy as np
import pytest
import torch

from lemas_tts_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,L", [(1, 2), (1, 64), (2, 301), (1, 939)])
def test_vocos_decode_vs_oracle(B, L):
    from lemas_tts_amd.engine import VocosEngine
    from oracle import lemas_oracle as O
    sd = synth.synth_vocos_state_dict(7)
    mel = torch.from_numpy(np.stack([synth.synth_cond_mel(100 + b, L, "vmel").T for b in range(B)]))
    ref = O.OracleVocos(sd).decode(mel)
    eng = VocosEngine(sd, device="cuda:0")
    wav = eng.decode(mel).cpu()
    assert wav.shape == ref.shape == (B, 256 * (L - 1))
    err = (wav - ref).abs().max().item()
    print(f"\n[vocos B={B} L={L}] max|err| {err:.3e}  |ref| max {ref.abs().max():.3e}")
    assert err < 1e-4 * max(1.0, ref.abs().max().item())
    half = eng.decode(mel, gain=0.5).cpu()
    np.testing.assert_allclose(half.numpy(), 0.5 * wav.numpy(), rtol=1e-6, atol=1e-7)
