"""CPU tier: host-side logic of the mirrored call surface that needs no GPU."""
import numpy as np
import torch

from lemas_tts_amd.infer.utils_infer import cross_fade_concat
from lemas_tts_amd.model.cfm import compute_sway_max, lens_to_mask, list_str_to_idx, time_grid
from oracle import lemas_oracle as O


def test_time_grid_matches_oracle_bitwise():
    for steps in (3, 4, 16, 32, 48):
        for coef in (None, 1, 3.0, 5, -1):
            assert torch.equal(time_grid(steps, coef), O.time_grid(steps, coef))
    assert abs(compute_sway_max(32) - 3.486) < 1e-3


def test_token_and_mask_helpers():
    vocab = {"a": 1, "b": 2, " ": 0}
    t = list_str_to_idx([["a", "b", "zz"], ["b"]], vocab)
    assert t.tolist() == [[1, 2, 0], [2, -1, -1]]          # unknown -> 0, pad -1 (model/utils.py:87-94)
    m = lens_to_mask(torch.tensor([2, 4]))
    assert m.tolist() == [[True, True, False, False], [True] * 4]


def test_cross_fade_matches_reference_formula():
    # utils_infer.py:581-617: linear fade over min(0.15 s, len(prev), len(next)) samples
    rng = np.random.default_rng(0)
    a, b, c = rng.standard_normal(6000), rng.standard_normal(5000), rng.standard_normal(100)
    out = cross_fade_concat([a, b, c], 0.15)
    n1 = int(0.15 * 24000)
    exp = np.concatenate([a[:-n1], a[-n1:] * np.linspace(1, 0, n1) + b[:n1] * np.linspace(0, 1, n1), b[n1:]])
    n2 = 100
    exp = np.concatenate([exp[:-n2], exp[-n2:] * np.linspace(1, 0, n2) + c[:n2] * np.linspace(0, 1, n2), c[n2:]])
    np.testing.assert_allclose(out, exp)
    np.testing.assert_array_equal(cross_fade_concat([a, b], 0.0), np.concatenate([a, b]))


def test_process_phone_list_rules():
    # api.py:252-276 -- language tags are consumed, phones get the current tag as prefix, '_' before punctuation drops
    from lemas_tts_amd.api import TTS
    t = TTS.__new__(TTS)
    t.langs = {"en": "en-us", "zh": "zh"}
    out = t.process_phone_list(["(en)", "h", "@", "_", ",", "_", "(zh)", "n", "i3", "."])
    assert out == ["(en)h", "(en)@", ",", "(zh)n", "(zh)i3", "."]


def test_tts_infer_frontend_dispatch_builds_the_reference_token_lists(monkeypatch):
    """api.py:199-216: phone frontend -> '|'-split phones with (cmn)->(zh); char frontend -> language tag + characters;
    one generation per '\\n'-separated line; separate_langs prefixes phones with the running language id."""
    import lemas_tts_amd.api as A
    seen = {}

    def fake_infer_process(ref_file, ref_text, gen_text, *a, **k):
        seen["ref"], seen["gen"] = ref_text, gen_text
        return None, 24000, None
    monkeypatch.setattr(A, "infer_process", fake_infer_process)
    tts = A.TTS.__new__(A.TTS)
    tts.ema_model = tts.vocoder = None
    tts.mel_spec_type, tts.device = "vocos", "cuda:0"
    tts.langs = {"cmn": "zh", "zh": "zh", "en": "en-us"}

    class _Phone:
        dtype = "phone"

        @staticmethod
        def text2phn(s):
            return "(cmn)|" + "|".join(s.strip(". ").split())
    tts.frontend = _Phone
    tts.infer(None, "a b", "c d\ne", seed=1)
    assert seen["ref"] == ["(zh)", "a", "b"] and seen["gen"] == [["(zh)", "c", "d"], ["(zh)", "e"]]
    tts.infer(None, "a b", "c", seed=1, separate_langs=True)
    assert seen["ref"] == ["(zh)a", "(zh)b"] and seen["gen"] == [["(zh)c"]]

    class _Char:
        dtype = "char"

        @staticmethod
        def text2norm(s):
            return "cmn", s.strip(". ")
    tts.frontend = _Char
    tts.infer(None, "ab", "cd\nef", seed=1)
    assert seen["ref"] == ["(zh)", "a", "b"] and seen["gen"] == [["(zh)", "c", "d"], ["(zh)", "e", "f"]]


def test_pad_rows_equals_pad_sequence():
    """cfm.pad_rows replaces torch.nn.utils.rnn.pad_sequence(batch_first=True) on the per-utterance host path (one [1875, 100] row through the
    library call cost 20 ms on 8 host threads): same tensors for ragged rows, other padding values, integer rows and the single-row case."""
    import torch
    from lemas_tts_amd.model.cfm import pad_rows
    g = torch.Generator().manual_seed(0)
    ragged = [torch.randn(n, 5, generator=g) for n in (7, 3, 11, 1)]
    assert torch.equal(pad_rows(ragged, 0), torch.nn.utils.rnn.pad_sequence(ragged, padding_value=0, batch_first=True))
    ids = [torch.arange(n) for n in (4, 9, 2)]
    out = pad_rows(ids, -1)
    assert out.dtype == torch.long and torch.equal(out, torch.nn.utils.rnn.pad_sequence(ids, padding_value=-1, batch_first=True))
    one = [torch.randn(6, 3, generator=g)]
    assert torch.equal(pad_rows(one, 0), torch.nn.utils.rnn.pad_sequence(one, batch_first=True)) and pad_rows(one, 0).shape == (1, 6, 3)
