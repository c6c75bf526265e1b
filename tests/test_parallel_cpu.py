"""CPU tier: the N>1 path (sharding, weight broadcast, gather) with two gloo processes.  The per-utterance compute
is stood in by the oracle on a depth-1 model -- tests may use the oracle; the product path never does."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch
from lemas_tts_amd.parallel import broadcast_state_dict, run_sharded, shard_utterances


def test_shard_balance_and_coverage():
    lens = [1125] * 64
    sh = shard_utterances(lens, 8)
    assert sorted(i for s in sh for i in s) == list(range(64)) and all(len(s) == 8 for s in sh)
    lens = [300, 1900, 800, 1200, 450, 1700, 950]
    sh = shard_utterances(lens, 2)
    assert sorted(i for s in sh for i in s) == list(range(7))
    cost = [sum(378_888_192.0 * lens[i] + 90_112.0 * lens[i] ** 2 for i in s) for s in sh]
    assert max(cost) / min(cost) < 1.25
    assert shard_utterances([], 4) == [[], [], [], []]
    assert shard_utterances([10], 4)[0] == [0]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import lemas_oracle as O
    arch, vocab = DiTArch(depth=1, conv_layers=1), 50
    sd0 = synth.synth_cfm_state_dict(arch, vocab, 5) if rank == 0 else None
    sd = broadcast_state_dict(sd0, arch, vocab, "cpu", dist)
    ref = synth.synth_cfm_state_dict(arch, vocab, 5)
    same = all(np.array_equal(sd[k], ref[k]) for k in ref) and set(sd) == set(ref)

    utts = [dict(seed=i, F=20 + 4 * i, N=48 + 8 * i) for i in range(5)]

    def fn(u):
        cond = torch.from_numpy(synth.synth_cond_mel(u["seed"], u["F"]))[None]
        text = torch.from_numpy(synth.synth_tokens(u["seed"], 10, vocab))[None]
        y0 = torch.from_numpy(synth.synth_noise(u["seed"], u["N"]))[None]
        out, _ = O.OracleCFM(sd, arch).sample(cond, text, u["N"], y0=y0, steps=2, cfg_strength=2.0, sway_sampling_coef=5)
        return out.numpy()

    res = run_sharded(utts, [u["N"] for u in utts], fn, dist)
    if rank == 0:
        single = [fn(u) for u in utts]
        exact = all(np.array_equal(a, b) for a, b in zip(res, single))
        q.put((same, exact, len(res)))
    else:
        q.put((same, None, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_run_matches_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(g[0] for g in got), "weight broadcast changed the tensors"
    r0 = [g for g in got if g[1] is not None][0]
    assert r0[1] and r0[2] == 5, "sharded results differ from the single-rank run"


def _job_worker(rank, world, port, q):
    """the sharded JOB's data movement (bench.py --job configs3_full): rank 0 owns every utterance, shards go out point to point,
    per-utterance results come back to rank 0 in input order"""
    from lemas_tts_amd.parallel import gather_to_rank0, scatter_from_rank0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    U, F_, nt = (12 if world < 8 else 64), 7, 5            # world 8: the real job's 64 utterances, 8 per rank, 7 peers in flight at once
    shards = shard_utterances([100] * U, world)
    per = len(shards[rank])
    packed = None
    if rank == 0:
        job = [(torch.full((F_, 3), float(i)), torch.full((nt,), i, dtype=torch.int64)) for i in range(U)]
        packed = [[torch.stack([job[i][0] for i in sh]).reshape(-1), torch.stack([job[i][1] for i in sh]).reshape(-1)] for sh in shards]
    like = [torch.empty(per * F_ * 3), torch.empty(per * nt, dtype=torch.int64)]
    fbuf, tbuf = scatter_from_rank0(packed, like, dist, "cpu")
    mine = fbuf.reshape(per, F_, 3)
    ok_in = all(float(mine[b, 0, 0]) == float(i) and int(tbuf.reshape(per, nt)[b, 0]) == i for b, i in enumerate(shards[rank]))
    res = mine.sum(dim=(1, 2))[:, None] * 2.0                  # "waveform" of each utterance: a function of its own input only
    got = gather_to_rank0(res, dist, "cpu")
    if rank == 0:
        host = torch.empty(U, 1)
        for r in range(world):
            host[torch.tensor(shards[r])] = got[r]
        q.put((ok_in, bool(torch.equal(host[:, 0], torch.arange(U).float() * F_ * 3 * 2.0))))
    else:
        q.put((ok_in, got is None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_job_scatter_and_gather_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(a and b for a, b in got), got


def _bf16_bcast_worker(rank, world, port, q):
    from lemas_tts_amd.parallel import block_gemm_weight, broadcast_bytes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arch, vocab = DiTArch(depth=2, conv_layers=1), 50
    ref = synth.synth_cfm_state_dict(arch, vocab, 9)
    sd = broadcast_state_dict(ref if rank == 0 else None, arch, vocab, "cpu", dist, block_weights_bf16=True)
    ok_order = list(sd) == list(ref)
    n_block = sum(block_gemm_weight(k) for k in ref)
    ok = n_block == 2 * 6
    for k, v in ref.items():
        if block_gemm_weight(k):       # travelled in bf16: exactly the value the bf16 step loop would have rounded to anyway
            ok &= bool(np.array_equal(sd[k], torch.from_numpy(v).to(torch.bfloat16).float().numpy()))
        else:                          # everything the fp32 hoists read is untouched
            ok &= bool(np.array_equal(sd[k], v))
    full, half = broadcast_bytes(arch, vocab), broadcast_bytes(arch, vocab, block_weights_bf16=True)
    ok &= full - half == 2 * sum(int(np.prod(v.shape)) for k, v in ref.items() if block_gemm_weight(k))
    q.put((ok_order, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_block_gemm_weights_can_travel_in_bf16():
    """broadcast_state_dict(block_weights_bf16=True): the DiT blocks' GEMM weights go out as a second, bf16 buffer (what the bf16 step loop
    rounds them to anyway: its results cannot change), every other tensor -- AdaLN linears, embeddings, biases -- stays fp32 bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(a and b for a, b in got), got
