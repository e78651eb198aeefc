"""N > 1 path on CPU: two gloo ranks shard a list of utterances, each 'analyses' its shard (the
features here are tagged dummies: the collective is what is under test) and rank 0 receives the
gather-v in rank order."""
import importlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sh = importlib.import_module("hts-train-world_amd.sharding")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lengths, q, handicap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = [sh.frame_count(n, 16000, 5.0) for n in lengths]
    shards = sh.lpt_shards(frames, world, sh.rank0_handicap(world) if handicap else None)
    mine = shards[rank]
    counts = [frames[i] for i in mine]
    total = sum(counts)
    # features tagged with (utterance id, frame index) so that the receiver can verify placement
    f0 = torch.cat([torch.full((frames[i],), float(i)) for i in mine]) if mine else torch.zeros(0)
    sp = torch.cat([torch.arange(frames[i], dtype=torch.float32).unsqueeze(1).repeat(1, 5) + 1000 * i for i in mine]) \
        if mine else torch.zeros(0, 5)
    res = sh.gather_features([f0, sp], counts, dst=0)
    if rank == 0:
        (g_f0, g_sp), all_counts = res
        order = [i for r in range(world) for i in shards[r]]
        assert [len(c) for c in all_counts] == [len(s) for s in shards]
        assert g_f0.shape[0] == sum(frames) and g_sp.shape == (sum(frames), 5)
        off = 0
        for i in order:
            assert torch.all(g_f0[off:off + frames[i]] == float(i))
            assert torch.all(g_sp[off:off + frames[i], 0] == torch.arange(frames[i]) + 1000 * i)
            off += frames[i]
        q.put("ok")
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_v():
    lengths = [32000, 80000, 48000, 127999, 64000, 16000, 100000]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"



def test_eight_rank_gather_v_with_rank0_handicap():
    """The node's full width on CPU: eight gloo ranks, the sweep's partition (rank 0 handicapped: it also moves
    everybody's features), ragged shards including short ones, gather-v to rank 0 in rank order."""
    rng = np.random.default_rng(3)
    lengths = [int(n) for n in rng.integers(16000, 128000, 61)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, lengths, q, True)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"
    frames = [sh.frame_count(n, 16000, 5.0) for n in lengths]
    shards = sh.lpt_shards(frames, 8, sh.rank0_handicap(8))
    loads = [sum(frames[i] for i in s) for s in shards]
    assert loads[0] < min(loads[1:])
