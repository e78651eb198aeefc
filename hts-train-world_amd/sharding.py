"""Utterance sharding across the GPUs of one node and the final feature gather.

The reference processes utterances one after another in a shell loop
(data/Makefile.in:125-242); they are independent, so the batch is partitioned across
ranks with no data-path collective.  The only exchange is the final gather of the
f0 / sp / ap arrays to rank 0 (which writes the feature files): a gather-v built from
grouped point-to-point send/recv (RCCL over xGMI when the backend is "nccl", gloo on
CPU in the tests).
"""
from __future__ import annotations

import numpy as np


def usable_cpus() -> int:
    """Host cores this process may actually use: the cgroup's CPU quota when there is one (a container on a large
    host sees all of the host's cores in os.cpu_count()), else the affinity mask, else os.cpu_count().  Sizing a
    thread pool by the host's 256 cores inside a 16-core share made the file writers three times slower."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:                                                    # cgroup v2: "max 100000" or "<quota> <period>"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                                # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def frame_count(n_samples: int, fs: int, frame_period: float) -> int:
    """GetSamplesForDIO (externs/WORLD_v2/src/dio.cpp:638-640)."""
    return int(1000.0 * n_samples / fs / frame_period) + 1


def lpt_shards(costs, world_size: int, handicap=None) -> list[list[int]]:
    """Longest-processing-time partition of utterance indices by cost (frame counts).

    handicap: per-rank cost factors >= 1 (a frame on rank r counts handicap[r] frames): the rank that also receives,
    copies and writes everybody's features (rank 0 of the sweep) analyses its frames more slowly and gets fewer of
    them, so that it is not the last to finish.  Deterministic: ties broken by index.  Every rank gets a (possibly
    empty) list."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    scale = [1.0] * world_size if handicap is None else [float(h) for h in handicap]
    assert len(scale) == world_size and all(h >= 1.0 for h in scale)
    load = [0.0] * world_size
    shards: list[list[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k] + costs[i] * scale[k], k))
        shards[r].append(i)
        load[r] += costs[i] * scale[r]
    for s in shards:
        s.sort()
    return shards


def rank0_handicap(world_size: int, factor=None) -> list[float]:
    """The cost factors of the sweep's partition (lpt_shards handicap): rank 0 > 1, the others 1.

    Rank 0 moves EVERY rank's features to the host while it analyses its own shard.  A device-to-host copy is a kernel
    on this platform, and the store-heavy Dio kernels that run beside one take several times their normal time
    (profiles/r04_d_kernel_trace.csv): with a copy always in flight a pass' analysis
    takes about 1.37 x.  On one rank the copies (5.5 ms per pass of the 1000-utterance corpus) cover 0.145 of the
    analysis (38 ms); on N ranks rank 0's analysis is N times shorter and the copies are not, so they cover
    min(1, 0.145 N) of it: factor = 1 + 0.37 min(1, 0.145 N) -- 1.11 / 1.21 / 1.37 at 2 / 4 / 8 ranks.
    WORLD_MI355_SWEEP_HANDICAP overrides rank 0's factor; one rank has none."""
    import os
    if world_size <= 1:
        return [1.0] * max(1, world_size)
    if factor is None:
        env = os.environ.get("WORLD_MI355_SWEEP_HANDICAP")
        factor = float(env) if env else 1.0 + 0.37 * min(1.0, 0.145 * world_size)
    return [max(1.0, factor)] + [1.0] * (world_size - 1)


def gather_features(tensors, frame_counts, dst: int = 0, group=None, all_counts=None):
    """Gather-v of per-rank feature slabs to ``dst``.

    tensors: list of torch tensors whose first dimension is this rank's total frame count
             (e.g. [f0 (T,), sp (T, bins), ap (T, bins)]).
    frame_counts: this rank's per-utterance frame counts (python ints).
    all_counts: every rank's frame_counts when the caller already knows them (a deterministic plan): skips the
             all_gather_object, which is a pickled host collective.
    Returns on dst: (list of gathered tensors in rank order, list of per-rank frame-count
    lists); elsewhere None.
    """
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if all_counts is not None:
        counts = [[int(c) for c in cs] for cs in all_counts]
        assert counts[rank] == [int(c) for c in frame_counts]
    else:
        counts = [None] * world
        dist.all_gather_object(counts, [int(c) for c in frame_counts], group=group)
    totals = [sum(c) for c in counts]
    if world == 1:
        return [t for t in tensors], counts
    ops, outs = [], []
    if rank == dst:
        offs = np.concatenate([[0], np.cumsum(totals)])
        for t in tensors:
            out = torch.empty((int(offs[-1]),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            out[offs[dst]:offs[dst + 1]].copy_(t)
            for r in range(world):
                if r != dst and totals[r] > 0:
                    ops.append(dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group))
            outs.append(out)
    elif totals[rank] > 0:
        for t in tensors:
            ops.append(dist.P2POp(dist.isend, t.contiguous(), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return (outs, counts) if rank == dst else None
