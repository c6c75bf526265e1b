"""Utterance-level data parallelism over the GPUs of one node.

The path shards naturally: one unit = one utterance (one ``process_batch`` call of the reference,
``lemas_tts/infer/utils_infer.py:506``; lines of ``gen_text`` are already independent, ``:572-579``).  One process
per GPU; the only collective is the one-off weight broadcast (RCCL over xGMI when the backend is ``nccl``); nothing
is exchanged inside the NFE loop.  The reference's only multi-GPU precedent is the same scheme at file level
(``uvr5/multiprocess_cuda_infer.py:404-420``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from .model.layout import cfm_param_shapes, vocos_param_shapes


def shard_utterances(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Longest-first deal of utterance indices to ranks (greedy least-loaded, cost ~ N^2 attention + N linear)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        n = float(lengths[i])
        load[r] += 378_888_192.0 * n + 90_112.0 * n * n
    return out


def _shapes(arch, vocab, vocos: bool, prosody: bool):
    return vocos_param_shapes() if vocos else cfm_param_shapes(arch, vocab, prosody)


def block_gemm_weight(name: str) -> bool:
    """True for the tensors the step loop only ever reads as bf16 MFMA operands: the DiT blocks' QKV / out-projection / FF weights
    (modules.py:452-454, :495, :349-350).  Everything else -- AdaLN linears, embeddings, biases, the vocoder -- feeds fp32 hoists."""
    if ".transformer_blocks." not in name or not name.endswith(".weight"):
        return False
    return any(k in name for k in (".attn.to_q.", ".attn.to_k.", ".attn.to_v.", ".attn.to_out.0.", ".ff.ff.0.0.", ".ff.ff.2."))


def broadcast_state_dict(sd: Optional[dict], arch, vocab, device, dist, *, vocos: bool = False, prosody: bool = False,
                         src: int = 0, block_weights_bf16: bool = False) -> dict:
    """Rank ``src`` holds ``sd`` (name -> fp32 array); every rank returns the same dict.  One flat fp32 buffer, one
    broadcast (~1.35 GB for the DiT: a single large collective suits per-link-bound xGMI rings).

    With a CUDA ``device`` (backend ``nccl`` = RCCL) the returned values are VIEWS of that flat device buffer: the engines
    load them with ``lemas_*_load_weight_device`` (device-to-device), so only rank ``src`` ever stages the weights through host
    memory -- that is what the broadcast buys over every rank reading the checkpoint.  With ``device="cpu"`` (the gloo rehearsal
    of the N > 1 path) numpy arrays come back, as from a checkpoint file.

    ``block_weights_bf16``: the DiT blocks' GEMM weights (0.74 of the 1.35 GB) travel as a SECOND flat buffer in bf16 -- the only form the
    bf16 step loop ever reads them in, so its results do not change by a bit (fp32 -> bf16 -> fp32 -> bf16 is the same rounding once).
    Not for the fp8 path, whose e4m3 quantiser starts from the fp32 masters: leave it off there (the default)."""
    shapes = _shapes(arch, vocab, vocos, prosody)
    dev = torch.device(device)
    groups = {False: [n for n in shapes if not (block_weights_bf16 and not vocos and block_gemm_weight(n))],
              True: [n for n in shapes if block_weights_bf16 and not vocos and block_gemm_weight(n)]}
    out = {}
    for half, names in groups.items():
        if not names:
            continue
        total = int(sum(int(np.prod(shapes[n])) for n in names))
        dtype = torch.bfloat16 if half else torch.float32
        flat = torch.empty(total, dtype=dtype, device=dev)
        if dist.get_rank() == src:
            host = np.empty(total, dtype=np.float32)
            off = 0
            for name in names:
                a = np.asarray(sd[name], dtype=np.float32).reshape(-1)
                assert a.size == int(np.prod(shapes[name])), name
                host[off: off + a.size] = a
                off += a.size
            flat.copy_(torch.from_numpy(host).to(dtype))          # ONE host-to-device copy per buffer on the source rank
        dist.broadcast(flat, src=src)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        off = 0
        for name in names:
            n = int(np.prod(shapes[name]))
            view = flat[off: off + n].reshape(tuple(shapes[name]))
            out[name] = view if dev.type == "cuda" else view.float().numpy()
            off += n
    return {n: out[n] for n in shapes}          # checkpoint order


def broadcast_bytes(arch, vocab, *, vocos: bool = False, prosody: bool = False, block_weights_bf16: bool = False) -> int:
    shapes = _shapes(arch, vocab, vocos, prosody)
    return int(sum(int(np.prod(s)) * (2 if (block_weights_bf16 and not vocos and block_gemm_weight(n)) else 4) for n, s in shapes.items()))


def gather_objects(obj, dist) -> Optional[list]:
    """Collect per-rank python results (waveforms are <= 1 MB per utterance) on rank 0."""
    world = dist.get_world_size()
    bucket = [None] * world if dist.get_rank() == 0 else None
    dist.gather_object(obj, bucket, dst=0)
    return bucket


def run_sharded(utterances: Sequence, lengths: Sequence[int], fn, dist) -> Optional[list]:
    """Apply ``fn(utterance)`` to this rank's shard; rank 0 gets all results back in input order."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = shard_utterances(lengths, world)[rank]
    local = [(i, fn(utterances[i])) for i in mine]
    allr = gather_objects(local, dist)
    if allr is None:
        return None
    out = [None] * len(utterances)
    for part in allr:
        for i, r in part:
            out[i] = r
    return out


def scatter_from_rank0(per_rank: Optional[Sequence[Sequence[torch.Tensor]]], like: Sequence[torch.Tensor], dist, device) -> List[torch.Tensor]:
    """Rank 0 holds ``per_rank[r]`` = the tensors of rank r's shard (e.g. [cond | y0 floats, token ids]); every rank returns its own, on
    ``device``.  ``like`` gives the shapes / dtypes a non-zero rank receives into.  Point to point: the shards differ per
    rank and nothing else needs them -- the job's only traffic besides the weight broadcast and the gather below (the
    reference's precedent hands each GPU worker its own file list, uvr5/multiprocess_cuda_infer.py:404-420).  NON-BLOCKING: rank 0 posts
    every send before it waits for any (``isend``), a receiver posts all its receives at once (``irecv``): with 7 peers the transfers run side
    by side over their own xGMI links instead of one after the other."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    dev = torch.device(device)
    if rank == 0:
        ops = []
        for r in range(1, world):
            for t in per_rank[r]:
                ops.append(dist.P2POp(dist.isend, t.to(dev).contiguous(), r))      # (the op keeps its buffer alive)
        reqs = dist.batch_isend_irecv(ops) if ops else []       # ONE group: RCCL launches every peer's transfer together
        mine = [t.to(dev) for t in per_rank[0]]
        for q in reqs:
            q.wait()
        return mine
    out = [torch.empty(tuple(t.shape), dtype=t.dtype, device=dev) for t in like]
    for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, t, 0) for t in out]):
        q.wait()
    return out


def gather_to_rank0(t: torch.Tensor, dist, device) -> Optional[List[torch.Tensor]]:
    """Every rank's equally shaped result tensor (its shard's waveforms) -> a list on rank 0 (host tensors), None elsewhere.  Rank 0 posts one
    receive per peer up front, the peers send as soon as they are done: a slow rank does not hold up the transfers of the others."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    buf = t.to(torch.device(device)).contiguous()
    if rank != 0:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, 0)]):
            q.wait()
        return None
    got = [torch.empty_like(buf) for _ in range(1, world)]
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, g, r) for r, g in zip(range(1, world), got)]) if got else []
    out = [buf.cpu()]
    for q in reqs:
        q.wait()
    out.extend(g.cpu() for g in got)
    return out


def gpu_numa_cpus(device_index: int):
    """(numa_node, cpu list) of the host NUMA node GPU ``device_index`` hangs off, from sysfs
    (``/sys/bus/pci/devices/<bdf>/{numa_node,local_cpulist}``); (None, None) when the platform does not say."""
    import os
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        node = int(open(os.path.join(base, "numa_node")).read().strip())
        cpus = []
        for part in open(os.path.join(base, "local_cpulist")).read().strip().split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            cpus.extend(range(int(lo), int(hi or lo) + 1))
        return (node if node >= 0 else None), (cpus or None)
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None, None


def pin_to_gpu_numa_node(device_index: int):
    """One process per GPU: keep this rank's host threads (launch path, pinned-buffer copies) on the cores of the GPU's own NUMA
    node, intersected with what the process may use.  Returns a small record for the report, or None when nothing was done."""
    import os
    node, cpus = gpu_numa_cpus(device_index)
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = os.sched_getaffinity(0)
    want = sorted(allowed.intersection(cpus))
    if not want:
        return None
    try:
        os.sched_setaffinity(0, want)
    except OSError:
        return None
    return {"numa_node": node, "cpus": len(want), "first_cpu": want[0], "last_cpu": want[-1]}
