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


def broadcast_state_dict(sd: Optional[dict], arch, vocab, device, dist, *, vocos: bool = False, prosody: bool = False,
                         src: int = 0) -> dict:
    """Rank ``src`` holds ``sd`` (name -> fp32 array); every rank returns the same dict.  One flat fp32 buffer, one
    broadcast (~1.35 GB for the DiT: a single large collective suits per-link-bound xGMI rings).

    With a CUDA ``device`` (backend ``nccl`` = RCCL) the returned values are VIEWS of that flat device buffer: the engines
    load them with ``lemas_*_load_weight_device`` (device-to-device), so only rank ``src`` ever stages the weights through host
    memory -- that is what the broadcast buys over every rank reading the checkpoint.  With ``device="cpu"`` (the gloo rehearsal
    of the N > 1 path) numpy arrays come back, as from a checkpoint file."""
    shapes = _shapes(arch, vocab, vocos, prosody)
    total = int(sum(int(np.prod(s)) for s in shapes.values()))
    dev = torch.device(device)
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    if dist.get_rank() == src:
        host = np.empty(total, dtype=np.float32)
        off = 0
        for name, shp in shapes.items():
            a = np.asarray(sd[name], dtype=np.float32).reshape(-1)
            assert a.size == int(np.prod(shp)), name
            host[off: off + a.size] = a
            off += a.size
        flat.copy_(torch.from_numpy(host))          # ONE host-to-device copy on the source rank
    dist.broadcast(flat, src=src)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    out, off = {}, 0
    for name, shp in shapes.items():
        n = int(np.prod(shp))
        view = flat[off: off + n].reshape(tuple(shp))
        out[name] = view if dev.type == "cuda" else view.numpy()
        off += n
    return out


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
    ``device``.  ``like`` gives the shapes / dtypes a non-zero rank receives into.  Point to point (send / recv): the shards differ per
    rank and nothing else needs them -- the job's only traffic besides the weight broadcast and the gather below (the
    reference's precedent hands each GPU worker its own file list, uvr5/multiprocess_cuda_infer.py:404-420)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    dev = torch.device(device)
    if rank == 0:
        for r in range(1, world):
            for t in per_rank[r]:
                dist.send(t.to(dev).contiguous(), dst=r)
        return [t.to(dev) for t in per_rank[0]]
    out = [torch.empty(tuple(t.shape), dtype=t.dtype, device=dev) for t in like]
    for t in out:
        dist.recv(t, src=0)
    return out


def gather_to_rank0(t: torch.Tensor, dist, device) -> Optional[List[torch.Tensor]]:
    """Every rank's equally shaped result tensor (its shard's waveforms) -> a list on rank 0 (host tensors), None elsewhere."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    buf = t.to(torch.device(device)).contiguous()
    if rank != 0:
        dist.send(buf, dst=0)
        return None
    out = [buf.cpu()]
    for r in range(1, world):
        got = torch.empty_like(buf)
        dist.recv(got, src=r)
        out.append(got.cpu())
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
