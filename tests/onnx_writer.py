"""Test helper: write a ConvTDFNet as an ONNX ModelProto (protobuf wire format, no onnx package) in the node layouts torch.onnx emits
for ``uvr5/lib_v5/mdxnet.py`` in eval mode, so that ``lemas_tts_amd/uvr5/onnx_weights.py`` has something to read.  NOT a pin: both sides
of that round trip are this repo's (see the reader's docstring)."""
import struct

import numpy as np


def _varint(x: int) -> bytes:
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _key(num, wt):
    return _varint((num << 3) | wt)


def _ld(num, payload: bytes) -> bytes:
    return _key(num, 2) + _varint(len(payload)) + payload


def _vi(num, x: int) -> bytes:
    return _key(num, 0) + _varint(x)


def tensor(name: str, arr: np.ndarray, raw: bool = True) -> bytes:
    arr = np.ascontiguousarray(arr)
    dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 7, np.dtype(np.float64): 11}[arr.dtype]
    out = b"".join(_vi(1, d) for d in arr.shape) + _vi(2, dt)
    if raw:
        out += _ld(9, arr.tobytes())
    elif dt == 1:
        out += _ld(4, arr.astype("<f4").tobytes())              # packed float_data
    elif dt == 7:
        out += _ld(7, b"".join(_varint(int(v)) for v in arr.reshape(-1)))
    else:
        out += _ld(10, arr.astype("<f8").tobytes())
    return out + _ld(8, name.encode())


def attr_ints(name, vals):
    return _ld(1, name.encode()) + b"".join(_vi(8, v) for v in vals) + _vi(20, 7)


def attr_int(name, v):
    return _ld(1, name.encode()) + _vi(3, v) + _vi(20, 2)


def attr_float(name, v):
    return _ld(1, name.encode()) + _key(2, 5) + struct.pack("<f", v) + _vi(20, 1)


def attr_tensor(name, t: bytes):
    return _ld(1, name.encode()) + _ld(5, t) + _vi(20, 4)


def node(op, inputs, outputs, attrs=(), name=""):
    out = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    return out + _ld(3, name.encode()) + _ld(4, op.encode()) + b"".join(_ld(5, a) for a in attrs)


def value_info(name, shape):
    dims = b"".join(_ld(1, _vi(1, d) if isinstance(d, int) else _ld(2, str(d).encode())) for d in shape)
    ttype = _vi(1, 1) + _ld(2, dims)
    return _ld(1, name.encode()) + _ld(2, _ld(1, ttype))


class Writer:
    def __init__(self, raw=True, constants_as_nodes=False):
        self.nodes, self.inits, self.raw, self.as_nodes, self.n = [], [], raw, constants_as_nodes, 0

    def const(self, name, arr):
        arr = np.asarray(arr)
        if self.as_nodes:
            self.nodes.append(node("Constant", [], [name], [attr_tensor("value", tensor("", arr, self.raw))]))
        else:
            self.inits.append(tensor(name, arr, self.raw))
        return name

    def fresh(self, stem="t"):
        self.n += 1
        return f"/{stem}_{self.n}"

    def op(self, op, inputs, attrs=()):
        out = self.fresh(op)
        self.nodes.append(node(op, inputs, [out], attrs, name=out))
        return out


def convtdfnet_onnx(path, arch, sd, fold_conv_bn=True, raw=True, constants_as_nodes=False, bn_eps=1e-5, dynamic_batch=True):
    """arch: oracle.mdx_oracle.MdxArch (BatchNorm variant); sd: its state dict."""
    W = Writer(raw, constants_as_nodes)
    cnt = [0]

    def bn_params(p):
        return sd[p + "weight"], sd[p + "bias"], sd[p + "running_mean"], sd[p + "running_var"]

    def conv_bn_relu(x, cp, bp, op="Conv", stride=1, k=1, pad=0, fold=True):
        w, b = sd[cp + "weight"].astype(np.float64), sd[cp + "bias"].astype(np.float64)
        attrs = [attr_ints("dilations", [1, 1]), attr_int("group", 1), attr_ints("kernel_shape", [k, k]), attr_ints("pads", [pad] * 4),
                 attr_ints("strides", [stride, stride])]
        if fold and op == "Conv":                                 # torch.onnx's eval-mode Conv + BatchNorm fusion: anonymous folded constants
            ga, be, mu, va = (v.astype(np.float64) for v in bn_params(bp))
            s = ga / np.sqrt(va + 1e-5)
            w, b = w * s[:, None, None, None], (b - mu) * s + be
            cnt[0] += 1
            y = W.op(op, [x, W.const(f"onnx::Conv_{900 + 2 * cnt[0]}", w.astype(np.float32)), W.const(f"onnx::Conv_{901 + 2 * cnt[0]}", b.astype(np.float32))], attrs)
        else:
            y = W.op(op, [x, W.const(cp + "weight", sd[cp + "weight"]), W.const(cp + "bias", sd[cp + "bias"])], attrs)
            ga, be, mu, va = bn_params(bp)
            va = (va.astype(np.float64) + 1e-5 - bn_eps).astype(np.float32)
            y = W.op("BatchNormalization", [y] + [W.const(bp + n, v) for n, v in zip(("weight", "bias", "running_mean", "running_var"), (ga, be, mu, va))],
                     [attr_float("epsilon", bn_eps), attr_float("momentum", 0.9)])
        return W.op("Relu", [y])

    def linear_bn_relu(x, lp, bp):
        cnt[0] += 1
        y = W.op("MatMul", [x, W.const(f"onnx::MatMul_{700 + cnt[0]}", np.ascontiguousarray(sd[lp + "weight"].T))])
        if lp + "bias" in sd:
            y = W.op("Add", [W.const(lp + "bias", sd[lp + "bias"]), y])
        ga, be, mu, va = bn_params(bp)
        y = W.op("BatchNormalization", [y] + [W.const(bp + n, v) for n, v in zip(("weight", "bias", "running_mean", "running_var"), (ga, be, mu, va))],
                 [attr_float("epsilon", 1e-5), attr_float("momentum", 0.9)])
        return W.op("Relu", [y])

    def block(x, p):
        for j in range(arch.l):
            x = conv_bn_relu(x, f"{p}tfc.H.{j}.0.", f"{p}tfc.H.{j}.1.", k=arch.k, pad=arch.k // 2, fold=fold_conv_bn)
        if arch.bn is None:
            return x
        y = linear_bn_relu(x, p + "tdf.0.", p + "tdf.1.")
        if arch.bn != 0:
            y = linear_bn_relu(y, p + "tdf.3.", p + "tdf.4.")
        return W.op("Add", [x, y])

    x = conv_bn_relu("input", "first_conv.0.", "first_conv.1.", fold=fold_conv_bn)
    x = W.op("Transpose", [x], [attr_ints("perm", [0, 1, 3, 2])])
    skips = []
    for i in range(arch.n):
        x = block(x, f"encoding_blocks.{i}.")
        skips.append(x)
        x = conv_bn_relu(x, f"ds.{i}.0.", f"ds.{i}.1.", stride=2, k=2, fold=fold_conv_bn)
    x = block(x, "bottleneck_block.")
    for i in range(arch.n):
        x = conv_bn_relu(x, f"us.{i}.0.", f"us.{i}.1.", op="ConvTranspose", stride=2, k=2, fold=False)
        x = W.op("Mul", [x, skips[-i - 1]])
        x = block(x, f"decoding_blocks.{i}.")
    x = W.op("Transpose", [x], [attr_ints("perm", [0, 1, 3, 2])])
    W.nodes.append(node("Conv", [x, W.const("final_conv.0.weight", sd["final_conv.0.weight"]), W.const("final_conv.0.bias", sd["final_conv.0.bias"])], ["output"],
                        [attr_ints("dilations", [1, 1]), attr_int("group", 1), attr_ints("kernel_shape", [1, 1]), attr_ints("pads", [0] * 4), attr_ints("strides", [1, 1])]))
    shape = ["batch_size" if dynamic_batch else 1, arch.dim_c, arch.dim_f, arch.dim_t]
    graph = b"".join(_ld(1, n) for n in W.nodes) + _ld(2, b"torch_jit") + b"".join(_ld(5, t) for t in W.inits) + \
        _ld(11, value_info("input", shape)) + _ld(12, value_info("output", shape))
    model = _vi(1, 7) + _ld(2, b"pytorch") + _ld(3, b"1.13.1") + _ld(7, graph) + _ld(8, _ld(1, b"") + _vi(2, 13))
    with open(path, "wb") as f:
        f.write(model)
