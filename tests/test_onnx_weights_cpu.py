"""The ONNX-file / checkpoint -> ConvTDFNet state-dict reader (lemas_tts_amd/uvr5/onnx_weights.py).  PARITY UNPINNED at this boundary: the
graphs are written by tests/onnx_writer.py in torch.onnx's layouts, not by torch.onnx (no onnx package in the image).  What IS checked
against pinned ground: the state dict the reader returns drives the oracle (pinned by the reference's class) to the same output as the
state dict the graph was written from."""
import os

import numpy as np
import pytest
import torch

from lemas_tts_amd.uvr5 import onnx_weights as OW
from oracle import mdx_oracle as MO
from onnx_writer import convtdfnet_onnx

ARCHS = {"mini": MO.MINI, "mini_wide": MO.MINI_WIDE, "mini_notdf": MO.MINI_NOTDF}


@pytest.mark.parametrize("name", sorted(ARCHS))
@pytest.mark.parametrize("fold,raw,as_nodes,eps", [(True, True, False, 1e-5), (False, False, True, 1e-3), (True, False, False, 1e-5)])
def test_reader_recovers_architecture_and_function(tmp_path, name, fold, raw, as_nodes, eps):
    arch = ARCHS[name]
    sd = MO.seeded_state_dict(arch, 4)
    path = str(tmp_path / "net.onnx")
    convtdfnet_onnx(path, arch, sd, fold_conv_bn=fold, raw=raw, constants_as_nodes=as_nodes, bn_eps=eps)
    got_arch, got_sd = OW.load_network_file(path, dim_t=None)
    want = dict(dim_c=arch.dim_c, dim_f=arch.dim_f, dim_t=arch.dim_t, num_blocks=arch.num_blocks, l=arch.l, g=arch.g, k=arch.k, bn=arch.bn,
                bias=arch.bias, optimizer="rmsprop")
    assert got_arch == want
    assert set(got_sd) == {k for k, _ in MO.schema(arch)}
    for k, shape in MO.schema(arch):
        assert got_sd[k].shape == shape, k
    x = MO.seeded_input(arch, 2, 8)
    ref = MO.MdxOracle(arch, sd).forward(x).numpy()
    out = MO.MdxOracle(MO.MdxArch(**got_arch), got_sd).forward(x).numpy()
    assert np.abs(out - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-5
    if not fold:                                                 # nothing folded: the named tensors come back verbatim (BN variance moved by eps)
        np.testing.assert_array_equal(got_sd["ds.0.0.weight"], sd["ds.0.0.weight"])
        np.testing.assert_allclose(got_sd["ds.0.1.running_var"], sd["ds.0.1.running_var"], rtol=1e-6)


def test_reader_refuses_what_is_not_a_convtdfnet(tmp_path):
    from onnx_writer import Writer, _ld, _vi, attr_ints, attr_int, node, value_info
    W = Writer()
    w = np.zeros((8, 4, 3, 3), np.float32)
    W.nodes.append(node("Conv", ["input", W.const("w", w), W.const("b", np.zeros(8, np.float32))], ["output"],
                        [attr_ints("kernel_shape", [3, 3]), attr_ints("strides", [1, 1]), attr_int("group", 1)]))
    graph = b"".join(_ld(1, n) for n in W.nodes) + b"".join(_ld(5, t) for t in W.inits) + _ld(11, value_info("input", [1, 4, 16, 8]))
    p = tmp_path / "other.onnx"
    p.write_bytes(_vi(1, 7) + _ld(7, graph))
    with pytest.raises(ValueError, match="kernel 3 stride 1"):
        OW.load_network_file(str(p))
    (tmp_path / "junk.onnx").write_bytes(b"\x08\x07")
    with pytest.raises(ValueError, match="no graph"):
        OW.load_network_file(str(tmp_path / "junk.onnx"))
    with pytest.raises(FileNotFoundError):
        OW.load_network_file(str(tmp_path / "absent.onnx"))


@pytest.mark.parametrize("ext", ["npz", "safetensors", "pt", "ckpt"])
def test_state_dict_files(tmp_path, ext):
    arch = MO.MINI
    sd = MO.seeded_state_dict(arch, 2)
    path = str(tmp_path / f"net.{ext}")
    if ext == "npz":
        np.savez(path, **sd)
    elif ext == "safetensors":
        from safetensors.numpy import save_file
        save_file(sd, path)
    elif ext == "pt":
        torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, path)
    else:                                                        # a Lightning-style checkpoint of the module
        torch.save({"state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "epoch": 3}, path)
    got_arch, got_sd = OW.load_network_file(path, dim_t=arch.dim_t)
    assert MO.MdxArch(**got_arch) == arch
    for k, v in sd.items():
        np.testing.assert_array_equal(got_sd[k], v)
    assert OW.arch_from_state_dict(MO.seeded_state_dict(MO.MINI_GN, 1), 16)["optimizer"] == "adamw"
    assert OW.arch_from_state_dict(MO.seeded_state_dict(MO.MINI_WIDE, 1), 8)["bn"] == 0
    with pytest.raises(ValueError, match="dim_f"):
        OW.arch_from_state_dict(MO.seeded_state_dict(MO.MINI_NOTDF, 1), 8)
    assert OW.arch_from_state_dict(MO.seeded_state_dict(MO.MINI_NOTDF, 1), 8, dim_f=16)["dim_f"] == 16
