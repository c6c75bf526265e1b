"""Weights of the MDX-Net from the files the reference ecosystem ships them in, WITHOUT onnx / onnxruntime (neither exists for this image).

The reference loads ``Kim_Vocal_1.onnx`` into an onnxruntime session (``uvr5/multiprocess_cuda_infer.py:225-238``).  That file is an
export of ``ConvTDFNet`` (``uvr5/lib_v5/mdxnet.py:36-127``), and the HIP engine (``lemas_mdx_*``) runs that architecture from the module's
state dict.  This module gets from the file to the state dict:

* ``read_onnx``: a minimal protobuf reader for the parts of an ONNX ModelProto that carry a network's structure -- graph nodes (op type,
  inputs, outputs, int / float / tensor attributes), initializers and Constant nodes (fp32 / fp64 / int64 tensors, ``raw_data`` or typed
  fields), the graph input's shape.  Field numbers are those of the published ``onnx.proto3``.
* ``convtdfnet_from_onnx``: walks the nodes in file order, keeps the parametric ones (Conv, ConvTranspose, MatMul / Gemm + its bias Add,
  BatchNormalization) and lays them over ConvTDFNet's forward order (first 1x1 conv; per block l 3x3 convs and the TDF linears; the
  stride-2 convs; the transposed convs; last 1x1 conv), inferring ``g, l, n, k, bn, bias, dim_f`` from the tensor shapes.  A Conv whose
  BatchNorm the exporter folded (torch.onnx does that in eval mode) gets an identity norm in the state dict.
* ``load_state_dict_file`` / ``arch_from_state_dict``: the same for a checkpoint of the module itself.

STATUS: PARITY UNPINNED at this boundary.  No ONNX file of the model is in the reference tree and nothing here can write one with
torch.onnx (it needs the onnx package), so the reader is tested against graphs emitted by this repo's own writer (tests/onnx_writer.py) in
the layouts torch.onnx is documented to produce; ``tools/first_contact.py`` is the check to run on the real file.  GroupNorm ('adamw')
exports are refused.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

BN_EPS = 1e-5


# ---- protobuf wire format --------------------------------------------------------------------------------------------------------
def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    val = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf: memoryview):
    """Yield (field number, wire type, value) over one message; length-delimited values come back as memoryviews."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f"protobuf wire type {wt} (groups) is not used by ONNX")
        yield num, wt, v


def _packed_varints(v, wt) -> List[int]:
    if wt == 0:
        return [v]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def _signed(x: int) -> int:
    return x - (1 << 64) if x >= (1 << 63) else x


_DTYPES = {1: np.float32, 11: np.float64, 7: np.int64, 6: np.int32, 10: np.float16}


def _tensor(buf: memoryview) -> Tuple[str, np.ndarray]:
    dims, dtype, name, raw = [], 1, "", None
    floats, doubles, int64s, int32s = [], [], [], []
    for num, wt, v in _fields(buf):
        if num == 1:
            dims += [_signed(x) for x in _packed_varints(v, wt)]
        elif num == 2:
            dtype = v
        elif num == 4:
            floats.append(np.frombuffer(bytes(v), dtype="<f4") if wt == 2 else np.frombuffer(v, dtype="<f4"))
        elif num == 5:
            int32s += [_signed(x) for x in _packed_varints(v, wt)]
        elif num == 7:
            int64s += [_signed(x) for x in _packed_varints(v, wt)]
        elif num == 8:
            name = bytes(v).decode()
        elif num == 9:
            raw = bytes(v)
        elif num == 10:
            doubles.append(np.frombuffer(bytes(v), dtype="<f8") if wt == 2 else np.frombuffer(v, dtype="<f8"))
        elif num in (13, 14) and (num == 13 or v == 1):
            raise NotImplementedError(f"tensor '{name}': external data files are not supported")
    if dtype not in _DTYPES:
        raise NotImplementedError(f"tensor '{name}': ONNX data type {dtype} is not supported")
    dt = np.dtype(_DTYPES[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, dtype=dt.newbyteorder("<"))
    elif floats:
        arr = np.concatenate(floats)
    elif doubles:
        arr = np.concatenate(doubles)
    elif int64s:
        arr = np.asarray(int64s, dtype=np.int64)
    elif int32s:
        arr = np.asarray(int32s, dtype=np.int32)
    else:
        arr = np.zeros(0, dtype=dt)
    return name, arr.astype(dt, copy=True).reshape(dims)          # own, writable memory (frombuffer views are read-only)


class Node:
    __slots__ = ("op", "name", "inputs", "outputs", "attrs")

    def __init__(self):
        self.op, self.name, self.inputs, self.outputs, self.attrs = "", "", [], [], {}


def _node(buf: memoryview) -> Node:
    n = Node()
    for num, wt, v in _fields(buf):
        if num == 1:
            n.inputs.append(bytes(v).decode())
        elif num == 2:
            n.outputs.append(bytes(v).decode())
        elif num == 3:
            n.name = bytes(v).decode()
        elif num == 4:
            n.op = bytes(v).decode()
        elif num == 5:
            aname, val = "", None
            ints, flts = [], []
            for an, aw, av in _fields(v):
                if an == 1:
                    aname = bytes(av).decode()
                elif an == 2:
                    val = struct.unpack("<f", av)[0]
                elif an == 3:
                    val = _signed(av)
                elif an == 4:
                    val = bytes(av)
                elif an == 5:
                    val = _tensor(av)[1]
                elif an == 7:
                    flts += list(np.frombuffer(bytes(av), dtype="<f4")) if aw == 2 else [struct.unpack("<f", av)[0]]
                elif an == 8:
                    ints += [_signed(x) for x in _packed_varints(av, aw)]
            n.attrs[aname] = val if val is not None else (ints if ints else flts)
    return n


class OnnxGraph:
    def __init__(self):
        self.nodes: List[Node] = []
        self.tensors: Dict[str, np.ndarray] = {}
        self.inputs: Dict[str, List[Optional[int]]] = {}


def _value_info(buf: memoryview) -> Tuple[str, List[Optional[int]]]:
    name, shape = "", []
    for num, _, v in _fields(buf):
        if num == 1:
            name = bytes(v).decode()
        elif num == 2:                                           # TypeProto
            for tn, _, tv in _fields(v):
                if tn == 1:                                      # tensor_type
                    for sn, _, sv in _fields(tv):
                        if sn == 2:                              # shape
                            for dn, _, dv in _fields(sv):
                                if dn == 1:                      # dim
                                    d = None
                                    for xn, _, xv in _fields(dv):
                                        if xn == 1:
                                            d = _signed(xv)
                                    shape.append(d)
    return name, shape


def read_onnx(path: str) -> OnnxGraph:
    with open(path, "rb") as f:
        data = memoryview(f.read())
    g = OnnxGraph()
    graph = None
    for num, wt, v in _fields(data):
        if num == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError(f"{path}: no graph in the ModelProto (not an ONNX model?)")
    for num, wt, v in _fields(graph):
        if num == 1:
            n = _node(v)
            if n.op == "Constant" and isinstance(n.attrs.get("value"), np.ndarray) and n.outputs:
                g.tensors[n.outputs[0]] = n.attrs["value"]
            else:
                g.nodes.append(n)
        elif num == 5:
            name, arr = _tensor(v)
            g.tensors[name] = arr
        elif num == 11:
            name, shape = _value_info(v)
            g.inputs[name] = shape
    for name in list(g.inputs):                                  # initializers listed as inputs (keep_initializers_as_inputs) are not inputs
        if name in g.tensors:
            del g.inputs[name]
    return g


# ---- ONNX graph -> ConvTDFNet state dict -----------------------------------------------------------------------------------------
def _parametric_ops(g: OnnxGraph) -> List[tuple]:
    ops = []
    for n in g.nodes:
        t = [g.tensors.get(i) for i in n.inputs]
        if n.op in ("Conv", "ConvTranspose"):
            if len(t) < 2 or t[1] is None:
                raise ValueError(f"{n.op} node '{n.name}': weight is not a constant")
            if n.attrs.get("group", 1) != 1 or any(d != 1 for d in n.attrs.get("dilations", [])) or any(p for p in n.attrs.get("output_padding", [])):
                raise NotImplementedError(f"{n.op} node '{n.name}': grouped / dilated / output-padded convolutions are not part of ConvTDFNet")
            w = np.asarray(t[1], dtype=np.float32)
            b = np.asarray(t[2], dtype=np.float32) if len(t) > 2 and t[2] is not None else np.zeros(w.shape[0 if n.op == "Conv" else 1], np.float32)
            strides = list(n.attrs.get("strides", [1, 1]))
            ops.append(("conv" if n.op == "Conv" else "convT", w, b, strides[0]))
        elif n.op == "MatMul":
            const = [x for x in t if x is not None]
            if len(const) == 1 and const[0].ndim == 2:
                ops.append(("linear", np.ascontiguousarray(np.asarray(const[0], dtype=np.float32).T)))       # [K, N] -> Linear's [N, K]
        elif n.op == "Gemm" and t[1] is not None:
            w = np.asarray(t[1], dtype=np.float32)
            ops.append(("linear", np.ascontiguousarray(w if n.attrs.get("transB", 0) else w.T)))
            if len(t) > 2 and t[2] is not None:
                ops.append(("bias", np.asarray(t[2], dtype=np.float32).reshape(-1)))
        elif n.op == "Add":
            const = [x for x in t if x is not None]
            if len(const) == 1 and ops and ops[-1][0] == "linear" and const[0].size == ops[-1][1].shape[0]:
                ops.append(("bias", np.asarray(const[0], dtype=np.float32).reshape(-1)))
        elif n.op == "BatchNormalization":
            if any(x is None for x in t[1:5]):
                raise ValueError(f"BatchNormalization node '{n.name}': parameters are not constants")
            ops.append(("bn",) + tuple(np.asarray(x, dtype=np.float32) for x in t[1:5]) + (float(n.attrs.get("epsilon", 1e-5)),))
        elif n.op in ("InstanceNormalization", "GroupNormalization"):
            raise NotImplementedError("GroupNorm ('adamw') exports are not supported by this reader: load the module's state dict instead")
    return ops


def convtdfnet_from_onnx(g: OnnxGraph, dim_t: Optional[int] = None):
    """-> (arch dict with the ConvTDFNet constructor arguments, state dict keyed like the reference module)."""
    ops = _parametric_ops(g)
    pos = 0
    sd: Dict[str, np.ndarray] = {}

    def peek(kind):
        return pos < len(ops) and ops[pos][0] == kind

    def norm(prefix, c):
        nonlocal pos
        if peek("bn"):
            _, scale, bias, mean, var, eps = ops[pos]
            pos += 1
            var = var + np.float32(eps - BN_EPS)                 # the engine normalises with eps 1e-5
        else:                                                    # folded by the exporter: identity
            scale, bias, mean = np.ones(c, np.float32), np.zeros(c, np.float32), np.zeros(c, np.float32)
            var = np.full(c, 1.0 - BN_EPS, np.float32)
        sd[prefix + "weight"], sd[prefix + "bias"], sd[prefix + "running_mean"], sd[prefix + "running_var"] = scale, bias, mean, var

    def conv(prefix, kind, want_stride=None, want_k=None):
        nonlocal pos
        if not peek(kind):
            raise ValueError(f"graph does not look like a ConvTDFNet: expected a {kind} for '{prefix}', found {ops[pos][0] if pos < len(ops) else 'the end'}")
        _, w, b, stride = ops[pos]
        if (want_stride and stride != want_stride) or (want_k and w.shape[-1] != want_k):
            raise ValueError(f"'{prefix}': kernel {w.shape[-1]} stride {stride} where ConvTDFNet has kernel {want_k} stride {want_stride}")
        pos += 1
        sd[prefix + "weight"], sd[prefix + "bias"] = w, b
        return w

    info = {"bn": None, "bias": False, "l": None, "k": None, "f": None}

    def block(prefix, c):
        nonlocal pos
        j = 0
        while peek("conv") and ops[pos][3] == 1 and ops[pos][1].shape[-1] > 1 and ops[pos][1].shape[:2] == (c, c):
            w = conv(f"{prefix}tfc.H.{j}.0.", "conv")
            norm(f"{prefix}tfc.H.{j}.1.", c)
            info["k"] = w.shape[-1]
            j += 1
        if j == 0 or (info["l"] is not None and info["l"] != j):
            raise ValueError(f"'{prefix}': {j} TFC convolutions where the first block has {info['l']}")
        info["l"] = j
        if not peek("linear"):
            return None
        idx = 0
        first = ops[pos][1]
        for _ in range(2):
            if not peek("linear"):
                break
            if idx == 3 and ops[pos][1].shape != first.shape[::-1]:
                break                                            # a Linear that is not the way back: not ours (cannot happen in ConvTDFNet)
            w = ops[pos][1]
            pos += 1
            sd[f"{prefix}tdf.{idx}.weight"] = w
            if peek("bias"):
                sd[f"{prefix}tdf.{idx}.bias"] = ops[pos][1]
                info["bias"] = True
                pos += 1
            norm(f"{prefix}tdf.{idx + 1}.", c)
            idx += 3
            if w.shape[0] == w.shape[1]:
                break                                            # bn == 0: a single square Linear
        h, f = first.shape
        info["bn"] = 0 if idx == 3 else f // h
        return f

    w = conv("first_conv.0.", "conv", 1, 1)
    gch, dim_c = w.shape[0], w.shape[1]
    norm("first_conv.1.", gch)
    c, i, fs = gch, 0, []
    while True:                                                  # encoder stages until a block is followed by a transposed convolution
        fs.append(block(f"encoding_blocks.{i}.", c))
        if peek("conv") and ops[pos][3] == 2:
            conv(f"ds.{i}.0.", "conv", 2, 2)
            norm(f"ds.{i}.1.", c + gch)
            c += gch
            i += 1
        else:
            break
    n = i
    for key in [k for k in sd if k.startswith(f"encoding_blocks.{n}.")]:      # the last block parsed was the bottleneck
        sd["bottleneck_block." + key.split(".", 2)[2]] = sd.pop(key)
    for i in range(n):
        conv(f"us.{i}.0.", "convT", 2, 2)
        norm(f"us.{i}.1.", c - gch)
        c -= gch
        block(f"decoding_blocks.{i}.", c)
    conv("final_conv.0.", "conv", 1, 1)
    if pos != len(ops):
        raise ValueError(f"graph does not look like a ConvTDFNet: {len(ops) - pos} parametric nodes after the last 1x1 convolution")
    shape = next(iter(g.inputs.values()), [])
    dim_f = fs[0] if fs[0] is not None else (shape[2] if len(shape) == 4 else None)
    if dim_t is None:
        dim_t = shape[3] if len(shape) == 4 and shape[3] else None
    if not dim_f or not dim_t:
        raise ValueError("dim_f / dim_t cannot be read from the graph (no TDF linears, no static input shape): pass them in the configuration")
    arch = dict(dim_c=dim_c, dim_f=int(dim_f), dim_t=int(dim_t), num_blocks=2 * n + 1, l=info["l"], g=gch, k=info["k"], bn=info["bn"],
                bias=info["bias"], optimizer="rmsprop")
    return arch, sd


# ---- the module's own checkpoint -------------------------------------------------------------------------------------------------
def arch_from_state_dict(sd: Dict[str, np.ndarray], dim_t: int, dim_f: Optional[int] = None) -> dict:
    """The ConvTDFNet constructor arguments a state dict of the module implies (shapes only)."""
    shp = {k: tuple(np.shape(v)) for k, v in sd.items()}
    gch, dim_c = shp["first_conv.0.weight"][:2]
    n = sum(1 for k in shp if k.startswith("ds.") and k.endswith(".0.weight"))
    l = sum(1 for k in shp if k.startswith("bottleneck_block.tfc.H.") and k.endswith(".0.weight"))
    k = shp["bottleneck_block.tfc.H.0.0.weight"][-1]
    bn, bias = None, "bottleneck_block.tdf.0.bias" in shp
    if "bottleneck_block.tdf.0.weight" in shp:
        h, fb = shp["bottleneck_block.tdf.0.weight"]
        bn = fb // h if "bottleneck_block.tdf.3.weight" in shp else 0
        dim_f = fb << n
    if not dim_f:
        raise ValueError("a network without TDF linears does not determine dim_f: pass it")
    opt = "rmsprop" if "first_conv.1.running_mean" in shp else "adamw"
    return dict(dim_c=dim_c, dim_f=int(dim_f), dim_t=int(dim_t), num_blocks=2 * n + 1, l=l, g=gch, k=k, bn=bn, bias=bias, optimizer=opt)


def load_state_dict_file(path: str) -> Dict[str, np.ndarray]:
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        return dict(np.load(path))
    if ext == ".safetensors":
        from safetensors.numpy import load_file
        return load_file(path)
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj:            # a Lightning checkpoint of the module
        obj = obj["state_dict"]
    return {k: v.detach().to(torch.float32).numpy() for k, v in obj.items() if hasattr(v, "detach")}


def load_network_file(path: str, dim_t: Optional[int] = None, dim_f: Optional[int] = None):
    """``*.onnx`` or a state-dict file -> (arch dict, state dict) for ``MdxEngine``."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    if path.lower().endswith(".onnx"):
        return convtdfnet_from_onnx(read_onnx(path), dim_t=dim_t)
    sd = load_state_dict_file(path)
    if dim_t is None:
        raise ValueError("a state-dict file does not carry dim_t: pass it")
    return arch_from_state_dict(sd, dim_t, dim_f), sd
