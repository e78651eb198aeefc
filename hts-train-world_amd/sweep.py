"""The recipe's feature-extraction sweep over a corpus, sharded across the GPUs of one node (BASELINE.json configs[3]).

The reference walks the corpus in a shell loop, one `analysis` process per utterance (data/Makefile.in:125-242,
call site :214).  Utterances are independent, so here

  1. every rank derives the same longest-processing-time partition of the utterance list from the frame counts
     alone (sharding.lpt_shards; a frame count needs the sample count only -- the wav header, not the samples);
  2. a rank loads and analyses ITS utterances only, in HBM-resident batches (Dio -> StoneMask -> CheapTrick -> D4C),
     and turns the features into the on-disk float32 form of the CLI (test/analysis.cpp:360-390 raw f0 / sp / ap,
     or :292-366 coded lf0 / mgc / bap when spec_dim is given);
  3. the only exchange: a gather-v of those float32 slabs to rank 0 (grouped send/recv: RCCL over xGMI with the
     "nccl" backend, every peer on its own link to rank 0; gloo on CPU in the tests);
  4. rank 0 copies them to pinned host memory and writes the files of every utterance (SURVEY.md 8(b) file contract).

Phases are timed separately (compute / gather / copy to host / file write) because the last two are rank 0's alone
and bound the strong scaling of the sweep, whatever the kernels do.

There is no CPU path: the analysis is libworld_mi355.so's.
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import capi, sharding, world as W

MAX_BATCH_FRAMES = 1_500_000          # about 12.3 GB of fp64 sp + ap at F = 1024; far inside 288 GB


def batches(order, frames, limit):
    """Consecutive groups of `order` whose frame counts stay within `limit` (a single larger utterance is its own group)."""
    cur, tot = [], 0
    for i in order:
        if cur and tot + frames[i] > limit:
            yield cur
            cur, tot = [], 0
        cur.append(i)
        tot += frames[i]
    if cur:
        yield cur


def tapered_batches(order, frames, rounds, limit, ratio=0.65):
    """`order` cut into `rounds` consecutive groups whose frame counts fall off geometrically (ratio per round), each
    within `limit`.  A pass is a pipeline -- round k + 1 is analysed while round k travels to the host and round k - 1
    is written -- so what no compute hides is the LAST round's transfer and file writes: the smaller that round, the
    shorter the tail (equal rounds: a quarter of the pass's copy + write time exposed at four rounds; tapered: a
    tenth).  Too small a round does not fill the GPU, hence the gentle ratio."""
    total = sum(frames[i] for i in order)
    w = [ratio ** k for k in range(rounds)]
    cum, acc = [], 0.0
    for x in w:
        acc += x
        cum.append(total * acc / sum(w))
    groups, cur, tot, done, k = [], [], 0, 0, 0
    for i in order:
        if cur and (tot + frames[i] > limit or (k < rounds - 1 and done + tot >= cum[k])):
            groups.append(cur)
            done += tot
            cur, tot = [], 0
            while k < rounds - 1 and done >= cum[k]:
                k += 1
        cur.append(i)
        tot += frames[i]
    if cur:
        groups.append(cur)
    return groups


class ShardedSweep:
    """One corpus, one sampling rate, `world` ranks.  Every rank builds the same plan; `load()` puts this rank's
    waveforms into HBM; `run()` is one pass: analyse, gather to rank 0, write."""

    def __init__(self, ctx, fs, frame_period, sample_counts, rank=0, world=1, spec_dim=0, ap_dim=24,
                 max_batch_frames=MAX_BATCH_FRAMES, backend="nccl", rounds=1, writers="rank0"):
        """rounds > 1 splits every shard into about that many batches, so that rank 0 copies and writes the
        features of one round while the next is being analysed.  writers="all": no gather -- every rank copies its
        own features to the host and writes its own files (what the rank-0 funnel of configs[3] costs is the
        difference between the two)."""
        self.ctx, self.fs, self.fp = ctx, int(fs), float(frame_period)
        self.rank, self.world, self.backend = rank, world, backend
        self.spec_dim, self.ap_dim = int(spec_dim), int(ap_dim)
        self.samples = [int(n) for n in sample_counts]
        self.frames = [sharding.frame_count(n, self.fs, self.fp) for n in self.samples]
        # rank 0 also moves everybody's features to the host and into files: fewer frames for it (sharding.rank0_handicap);
        # when every rank writes its own shard there is no such rank
        self.handicap = sharding.rank0_handicap(world) if writers == "rank0" else [1.0] * world
        self.shards = sharding.lpt_shards(self.frames, world, self.handicap)
        self.writers = writers
        # batch plan of EVERY rank (rank 0 needs the layout of what it receives): longest first inside a shard; with
        # several rounds the rounds taper off (tapered_batches)
        by_len = lambda s: sorted(s, key=lambda i: (-self.frames[i], i))
        if rounds > 1:
            ratio = float(os.environ.get("WORLD_MI355_SWEEP_TAPER", "0.65"))
            self.plan = [tapered_batches(by_len(s), self.frames, rounds, max_batch_frames, ratio) for s in self.shards]
        else:
            self.plan = [list(batches(by_len(s), self.frames, max_batch_frames)) for s in self.shards]
        self.rounds = max((len(p) for p in self.plan), default=0)
        self.params = W.default_params(self.fs, self.fp)
        self.loaded = []                 # [(WorldBatch, x on device)] for this rank's batches
        self._pinned = {}                # rank 0's host staging of the gathered slabs
        self._items = {}                 # round -> (sink, pinned pointers, the native writer's list for that round)
        self._write_busy = 0.0
        self.total_frames = sum(self.frames)
        self.my_frames = sum(self.frames[i] for i in self.shards[rank])

    # ---- input ------------------------------------------------------------------------------------------------
    def load(self, waveform_of, io_threads=8):
        """waveform_of(i) -> float64 samples of utterance i.  Called for this rank's utterances only."""
        import torch
        self.close()
        with ThreadPoolExecutor(io_threads) as pool:
            for group in self.plan[self.rank]:
                xs = list(pool.map(waveform_of, group))
                for i, x in zip(group, xs):
                    if len(x) != self.samples[i]:
                        raise ValueError("utterance %d has %d samples, the plan says %d" % (i, len(x), self.samples[i]))
                b = W.WorldBatch(self.ctx, self.params, x_lengths=[len(x) for x in xs])
                x = torch.from_numpy(np.concatenate(xs)).pin_memory().cuda(non_blocking=True)
                self.loaded.append((b, x))
        torch.cuda.synchronize()

    def close(self):
        for b, _ in self.loaded:
            b.close()
        self.loaded = []
        self._items = {}                 # the write lists hold views of the pinned slabs and their sinks

    # ---- one pass ---------------------------------------------------------------------------------------------
    def _features(self, b, x):
        t, f0, sp, ap = b.analyze(x)
        if self.spec_dim:
            return list(b.recipe_features(f0, sp, ap, self.spec_dim, self.ap_dim))          # analysis.cpp:292-366
        return [f0.float(), sp.float(), ap.float()]                                          # analysis.cpp:360-390

    def run(self, sink=None, io_threads=None):
        """One pass over the corpus.  sink(i, f0, sp, ap) is called on rank 0 (on every rank for its own utterances
        when writers == "all") for every utterance i with float32 numpy views (from a thread pool; the arrays stay
        valid until run() returns).  Returns the phase times in seconds on this rank: compute, gather, to_host, write
        (the last two are zero off rank 0) and, from the pipelined form, `wall`.

        With the features on the GPU the pass is a pipeline over the rounds: round k+1 is analysed on the compute
        stream while round k's slabs are gathered and copied to pinned host memory on a second stream and round
        k-1's files are being written by the pool; the phases are then busy times (HIP events per stream, the pool's
        first submit to last completion) that overlap inside `wall`.  Over gloo (the rehearsal of N ranks on one GPU)
        the same pipeline runs with the two transfer steps swapped: every rank copies its round to pinned host
        memory on the second stream and the gather-v of round k is the host collective issued while round k+1 is
        on the GPU."""
        import torch
        if io_threads is None:
            io_threads = max(4, min(64, 2 * sharding.usable_cpus()))     # page-cache fills: two per usable core
        on_gpu = torch.cuda.is_available() and self.ctx is not None
        if on_gpu:
            return self._run_pipelined(sink, io_threads)
        return self._run_serial(sink, io_threads)

    def _submit(self, pool, sink, host, groups, pending, io_threads=16, key=None):
        """Hand one round's utterances to the writers.  With a DirSink the round is ONE job: the list of (path, row
        slice) pairs -- built once per (round, sink) and kept, the pinned slabs are the same every pass -- goes to
        the native writer; any other sink is called per utterance from the pool."""
        native = hasattr(sink, "paths")
        cached = self._items.get(key) if (native and key is not None) else None
        # a hit needs the SAME sink object (the entry holds it, so its identity cannot be handed to another sink
        # while the entry lives) and the same pinned slabs
        if cached is not None and cached[0] is sink and cached[1] == tuple(h.data_ptr() for h in host):
            pending.append(pool.submit(self._timed_write, cached[2], io_threads))
            return
        arrs = [h.numpy() for h in host]
        off = 0
        items = []
        for g in groups:                              # rank order, then batch order: the gather's layout
            for i in g:
                e = off + self.frames[i]
                if native:                            # row slices of contiguous slabs: contiguous themselves
                    items.extend(zip(sink.paths(i), (arrs[0][off:e], arrs[1][off:e], arrs[2][off:e])))
                else:
                    pending.append(pool.submit(sink, i, arrs[0][off:e], arrs[1][off:e], arrs[2][off:e]))
                off = e
        if items:
            if key is not None:
                self._items[key] = (sink, tuple(h.data_ptr() for h in host), items)
            pending.append(pool.submit(self._timed_write, items, io_threads))

    def _timed_write(self, items, io_threads):
        t0 = time.perf_counter()
        W.write_files(items, io_threads)
        self._write_busy += time.perf_counter() - t0

    def _run_pipelined(self, sink, io_threads):
        import torch
        import torch.distributed as dist
        comp = torch.cuda.current_stream()
        if getattr(self, "_xfer", None) is None:
            self._xfer = torch.cuda.Stream()
        xfer = self._xfer
        me_writes = self.rank == 0 or self.writers == "all"
        host_gather = self.world > 1 and self.backend == "gloo" and self.writers != "all"
        gather_busy = [0.0]
        round_counts = []
        ev = lambda: torch.cuda.Event(enable_timing=True)
        marks = []                                     # per round: (compute start, compute end, transfer start, gathered, on host)
        hosts, round_groups, keep = [], [], []
        pending = []
        t_first_submit = [None]
        self._write_busy = 0.0
        t0 = time.perf_counter()
        with ThreadPoolExecutor(io_threads) as pool:
            dbg = [] if os.environ.get("WM_SWEEP_TIMELINE") else None

            def drain(k):
                """round k is on the host: hand its utterances to the writers"""
                if dbg is not None:
                    dbg.append(("drain%d wait" % k, time.perf_counter() - t0))
                marks[k][4].synchronize()
                if dbg is not None:
                    dbg.append(("drain%d on host" % k, time.perf_counter() - t0))
                if host_gather:                        # every rank, every round, in the same order: a collective
                    tg = time.perf_counter()
                    res = sharding.gather_features(hosts[k], round_counts[k][self.rank], dst=0,
                                                   all_counts=round_counts[k])
                    gather_busy[0] += time.perf_counter() - tg
                    hosts[k] = res[0] if self.rank == 0 else None
                if sink is not None and hosts[k] is not None:
                    if t_first_submit[0] is None:
                        t_first_submit[0] = time.perf_counter()
                    self._submit(pool, sink, hosts[k], round_groups[k], pending, io_threads, key=k)

            for k in range(self.rounds):
                c0, c1, x0, x1, x2 = ev(), ev(), ev(), ev(), ev()
                c0.record(comp)
                feats = self._features(*self.loaded[k]) if k < len(self.loaded) else None
                c1.record(comp)
                groups = [p[k] if k < len(p) else [] for p in self.plan]
                counts = [[self.frames[i] for i in g] for g in groups]
                host = None
                with torch.cuda.stream(xfer):
                    xfer.wait_event(c1)
                    x0.record(xfer)
                    if self.writers == "all" or self.world == 1:
                        if self.writers == "all":
                            groups = [g if r == self.rank else [] for r, g in enumerate(groups)]
                        got = feats
                    elif host_gather:
                        if feats is None:
                            feats = self._empty()
                        got = feats                    # down to the host first; drain() gathers over gloo
                    else:
                        if feats is None:
                            feats = self._empty()
                        res = sharding.gather_features(feats, counts[self.rank], dst=0, all_counts=counts)
                        got = res[0] if self.rank == 0 else None
                    x1.record(xfer)
                    if feats is not None:
                        for v in feats:
                            v.record_stream(xfer)
                        keep.append(feats)
                    if (me_writes or host_gather) and got is not None:
                        host = []
                        for j, v in enumerate(got):
                            # pinned staging, allocated once per (round, array) and reused by later passes
                            h = self._pinned.get((k, j))
                            if h is None or h.shape != v.shape:
                                h = self._pinned[(k, j)] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                            h.copy_(v, non_blocking=True)
                            host.append(h)
                        keep.append(got)
                    x2.record(xfer)
                marks.append((c0, c1, x0, x1, x2))
                hosts.append(host)
                round_groups.append(groups)
                round_counts.append(counts)
                if k >= 1:
                    drain(k - 1)                      # while round k runs on the GPU
            if self.rounds:
                drain(self.rounds - 1)
            if dbg is not None:
                dbg.append(("all submitted", time.perf_counter() - t0))
            for p in pending:
                p.result()
            t_end = time.perf_counter()
            if dbg is not None:
                dbg.append(("writers done", t_end - t0))
        torch.cuda.synchronize()
        ph = {"compute": 0.0, "gather": 0.0, "to_host": 0.0, "write": 0.0}
        for c0, c1, x0, x1, x2 in marks:
            ph["compute"] += c0.elapsed_time(c1) * 1e-3
            ph["gather"] += x0.elapsed_time(x1) * 1e-3
            ph["to_host"] += x1.elapsed_time(x2) * 1e-3
        if host_gather:
            ph["gather"] = gather_busy[0]
        if t_first_submit[0] is not None:
            # time inside the native writer when the sink is a DirSink (jobs of different rounds may overlap), else the
            # span from the first submit to the last completion
            ph["write"] = self._write_busy if self._write_busy > 0 else t_end - t_first_submit[0]
        ph["wall"] = t_end - t0
        if dbg is not None:
            base = marks[0][0]
            ev_ms = [(base.elapsed_time(m[1]), base.elapsed_time(m[2]), base.elapsed_time(m[4])) for m in marks]
            import sys
            print("  timeline ms: " + ", ".join("%s %.2f" % (n, v * 1e3) for n, v in dbg), file=sys.stderr)
            print("  rounds (compute end, copy start, on host; ms after the first kernel): "
                  + "; ".join("%.2f %.2f %.2f" % e for e in ev_ms), file=sys.stderr)
        if self.world > 1 and dist.is_initialized():
            dist.barrier()
        return ph

    def _run_serial(self, sink, io_threads):
        """The same pass one phase after another (CPU tensors: the gloo rehearsals and tests)."""
        import torch
        import torch.distributed as dist
        ph = {"compute": 0.0, "gather": 0.0, "to_host": 0.0, "write": 0.0}
        sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
        t_pass = time.perf_counter()
        with ThreadPoolExecutor(io_threads) as pool:
            pending = []
            keep = []
            for k in range(self.rounds):
                t0 = time.perf_counter()
                feats = None
                if k < len(self.loaded):
                    feats = self._features(*self.loaded[k])
                sync()
                t1 = time.perf_counter()
                ph["compute"] += t1 - t0
                # what each rank contributes in this round, known everywhere from the plan
                groups = [p[k] if k < len(p) else [] for p in self.plan]
                counts = [[self.frames[i] for i in g] for g in groups]
                if self.writers == "all":
                    groups = [g if r == self.rank else [] for r, g in enumerate(groups)]
                    got = feats
                elif self.world > 1:
                    if feats is None:
                        feats = self._empty()
                    if self.backend == "gloo":
                        feats = [v.cpu() for v in feats]
                    res = sharding.gather_features(feats, counts[self.rank], dst=0, all_counts=counts)
                    if self.backend != "gloo":
                        sync()
                    got = res[0] if self.rank == 0 else None
                else:
                    got = feats
                t2 = time.perf_counter()
                ph["gather"] += t2 - t1
                if (self.rank == 0 or self.writers == "all") and got is not None:
                    host = []
                    for j, v in enumerate(got):
                        if v.is_cuda:
                            h = self._pinned.get((k, j))
                            if h is None or h.shape != v.shape:
                                h = self._pinned[(k, j)] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                            h.copy_(v, non_blocking=True)
                        else:
                            h = v
                        host.append(h)
                    sync()
                    t3 = time.perf_counter()
                    ph["to_host"] += t3 - t2
                    if sink is not None:
                        keep.append(host)
                        self._submit(pool, sink, host, groups, pending, io_threads)
            t4 = time.perf_counter()
            for p in pending:
                p.result()
            ph["write"] = time.perf_counter() - t4 if pending else 0.0
        ph["wall"] = time.perf_counter() - t_pass
        if self.world > 1 and dist.is_initialized():
            dist.barrier()
        return ph

    def _empty(self):
        import torch
        bins = (int(self.params.fft_size) or capi.cheaptrick_fft_size(self.fs)) // 2 + 1
        cols = (self.spec_dim, self.ap_dim) if self.spec_dim else (bins, bins)
        return [torch.empty(0, dtype=torch.float32, device="cuda"),
                torch.empty(0, cols[0], dtype=torch.float32, device="cuda"),
                torch.empty(0, cols[1], dtype=torch.float32, device="cuda")]


def file_sink(paths):
    """sink for ShardedSweep.run(): paths[i] = (f0_out, sp_out, ap_out); raw float32, as the CLI writes them."""
    def put(i, f0, sp, ap):
        for path, arr in zip(paths[i], (f0, sp, ap)):
            arr.tofile(path)                      # row slices of a contiguous slab: no copy
    return put


class DirSink:
    """sink for ShardedSweep.run(): utterance i goes to out_dir/<name>/utt%05d.<name> for the three names given --
    one directory per feature, as the recipe lays its files out (data/Makefile.in:214: lf0/$base.lf0, mgc/$base.mgc,
    bap/$base.bap; creating files in ONE directory serialises on that directory's lock).  Callable like any sink;
    run() recognises `paths` and hands a whole round to the native writer instead (one library call, plain threads:
    WorldMi355WriteFiles), which is what keeps 3 000 files per pass off the interpreter lock."""

    def __init__(self, out_dir, names=("f0", "sp", "ap")):
        self.out_dir, self.names = out_dir, tuple(names)
        for n in self.names:
            os.makedirs(os.path.join(out_dir, n), exist_ok=True)

    def paths(self, i):
        return [os.path.join(self.out_dir, n, "utt%05d.%s" % (i, n)) for n in self.names]

    def __call__(self, i, f0, sp, ap):
        for path, arr in zip(self.paths(i), (f0, sp, ap)):
            arr.tofile(path)


def dir_sink(out_dir, names=("f0", "sp", "ap")):
    return DirSink(out_dir, names)
