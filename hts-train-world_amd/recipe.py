"""Batched stand-in for the recipe's per-utterance loops around the two CLIs.

The reference runs one process per utterance (data/Makefile.in:206-216):

    analysis  WAV  LF0  MGC  BAP  [frame_period [fft_size [spec_dim [ap_dim]]]]   (test/analysis.cpp:243-390)
    synth     LF0  MGC  BAP  WAV  frame_period fft_size fs [spec_dim [ap_dim]]    (test/synth.cpp:124-262)

Here a whole file list is read, packed into HBM-resident batches, analysed or synthesised in one set of
launches per batch, and written back in the CLIs' own file formats (SURVEY.md 8(b) "File contract"):

    wav in      16-bit mono PCM, x = s / 2^(nbit-1)                       (test/audioio.cpp:236-249)
    f0 / lf0    float32 [T]            Hz, or log Hz with 0 for unvoiced when spec_dim != 0
    sp / mgc    float32 [T][F/2+1]     or [T][spec_dim] coded
    ap / bap    float32 [T][F/2+1]     or [T][ap_dim] coded
    wav out     16-bit mono PCM, s = clip(int(y * 32767))                 (test/audioio.cpp:160-167)

Settings are the CLI's: Dio(71-800 Hz, speed 1, allowed_range 0.1) + StoneMask, CheapTrick(q1 -0.15),
D4C(threshold 0) (analysis.cpp:93-203).  With several ranks (torch.distributed initialised, or WORLD_SIZE in
the environment) the list is sharded by frame count (sharding.lpt_shards; the counts come from the wav headers, a rank decodes only
its own utterances) and every rank writes its own files, or -- `--gather`, BASELINE.json configs[3] -- the float32
features travel to rank 0, which writes them all.

    python -m hts-train-world_amd.recipe analysis --scp jobs.txt --frame-period 5 --fft-size 2048 \\
           --spec-dim 50 --ap-dim 25          # jobs.txt: one "wav f0 sp ap" per line
    python -m hts-train-world_amd.recipe synth --scp jobs.txt --frame-period 5 --fft-size 2048 --fs 48000 \\
           --spec-dim 50 --ap-dim 25          # jobs.txt: one "f0 sp ap wav" per line

cmp_files() is the stage after it (data/Makefile.in:244-323: window.pl per stream, merge, addhtkheader.pl).

There is no CPU path: without a HIP device the library call fails.
"""
from __future__ import annotations

import argparse
import os
import struct
import sys
import wave
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import capi, sharding, world as W

MAX_BATCH_FRAMES = 1_500_000          # about 12.3 GB of fp64 sp + ap at F = 1024; far inside 288 GB


# ---- file formats -----------------------------------------------------------------------------------------
def read_wav(path):
    """x in [-1, 1) and fs, as test/audioio.cpp wavread() hands them over (little-endian two's complement
    of nbit bits over 2^(nbit-1); the first channel layout is taken as it is: the CLIs assume mono)."""
    with wave.open(str(path), "rb") as w:
        nbytes, fs, n = w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if nbytes == 2:
        s = np.frombuffer(raw, dtype="<i2").astype(np.float64)
    else:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, nbytes).astype(np.int64)
        s = sum(b[:, k] << (8 * k) for k in range(nbytes))
        s = np.where(s >= 1 << (8 * nbytes - 1), s - (1 << (8 * nbytes)), s).astype(np.float64)
    return s / float(1 << (8 * nbytes - 1)), fs


def write_wav(path, y, fs):
    """test/audioio.cpp wavwrite(): 44-byte header, int16 = clip(trunc(y * 32767))."""
    s = np.clip(np.trunc(np.asarray(y, dtype=np.float64) * 32767.0), -32768, 32767).astype("<i2")
    n = len(s)
    head = b"RIFF" + struct.pack("<I", 36 + n * 2) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, fs, fs * 2, 2, 16)
    with open(path, "wb") as f:
        f.write(head + b"data" + struct.pack("<I", n * 2))
        f.write(s.tobytes())


def _f32(path, cols=None):
    a = np.fromfile(path, dtype=np.float32)
    return a if cols is None else a.reshape(-1, cols)


# ---- batching ---------------------------------------------------------------------------------------------
def _my_share(costs):
    """Indices of this rank's jobs (all of them on a single process)."""
    rank, world = _rank_world()
    if world == 1:
        return list(range(len(costs)))
    return sharding.lpt_shards(costs, world)[rank]


def _own_context():
    """A context on this rank's GPU (LOCAL_RANK of a torchrun launch, else the current device)."""
    import torch
    local = os.environ.get("LOCAL_RANK")
    if local is not None:
        torch.cuda.set_device(int(local))
        return W.Context(device=int(local))
    return W.Context()


def _batches(order, frames, limit):
    cur, tot = [], 0
    for i in order:
        if cur and tot + frames[i] > limit:
            yield cur
            cur, tot = [], 0
        cur.append(i)
        tot += frames[i]
    if cur:
        yield cur


def wav_header(path):
    """(sample count, sampling rate) from the header alone: what the partition needs (no samples are decoded)."""
    with wave.open(str(path), "rb") as w:
        return w.getnframes(), w.getframerate()


def outputs_complete(job, n_frames, fs, spec_dim=0, ap_dim=24):
    """True when the three feature files of `job` = (wav, f0_out, sp_out, ap_out) exist with exactly the sizes the
    analysis of this wav writes (float32: n_frames, n_frames x width, n_frames x width; width = CheapTrick's bins, or
    spec_dim / ap_dim in the recipe's coded form, analysis.cpp:292-390).  The test behind `resume`: a run that was cut
    short left its last files missing or short (the native writer creates, fills and closes file after file)."""
    bins = capi.cheaptrick_fft_size(fs) // 2 + 1
    want = (4 * n_frames, 4 * n_frames * (spec_dim if spec_dim else bins), 4 * n_frames * (ap_dim if spec_dim else bins))
    try:
        return all(os.path.getsize(str(p)) == w for p, w in zip(job[1:4], want))
    except OSError:
        return False


def analysis_files(jobs, frame_period=5.0, fft_size=0, spec_dim=0, ap_dim=24, ctx=None,
                   max_batch_frames=MAX_BATCH_FRAMES, io_threads=8, gather=False, resume=False):
    """jobs: [(wav, f0_out, sp_out, ap_out)].  Writes what `analysis wav f0 sp ap frame_period fft_size
    [spec_dim [ap_dim]]` writes for every job; returns the number of frames analysed by this rank.

    Every rank reads the wav HEADERS of the whole list (the partition needs the frame counts), but decodes only the
    utterances of its own shard, one batch at a time.  gather=False: every rank writes the files of its shard.
    gather=True (BASELINE.json configs[3]): the float32 slabs go to rank 0 (sweep.ShardedSweep), which writes all files.
    resume=True (every rank writes its shard only): a rank skips the jobs of ITS shard whose outputs are complete
    (outputs_complete) -- the partition is made on the whole list first, so ranks that start at different times agree on
    it whatever is on disk; the reference's recipe re-runs every utterance (data/Makefile.in:206-216)."""
    import torch
    jobs = list(jobs)
    with ThreadPoolExecutor(io_threads) as pool:
        heads = list(pool.map(lambda j: wav_header(j[0]), jobs))
        frames = [sharding.frame_count(n, fs, frame_period) for n, fs in heads]
        for fs in sorted({h[1] for h in heads}):
            own_size = capi.cheaptrick_fft_size(fs)                    # GetFFTSizeForCheapTrick at the 71 Hz floor
            if fft_size not in (0, own_size):
                # analysis.cpp:157-179 sizes the rows by argv[6] but CheapTrick still runs at its own default
                # size for fs: any other value makes the reference write past or short of its rows
                raise ValueError("fft_size %d is not CheapTrick's size for %d Hz (%d)" % (fft_size, fs, own_size))
        own_ctx = ctx is None
        ctx = ctx or _own_context()
        if gather and resume:
            raise ValueError("resume goes with every rank writing its own shard (gather=False)")
        if gather:
            done = _analysis_gathered(jobs, heads, frame_period, spec_dim, ap_dim, ctx, max_batch_frames, io_threads)
            if own_ctx:
                ctx.close()
            return done
        mine = _my_share(frames)
        if resume:
            mine = [i for i in mine if not outputs_complete(jobs[i], frames[i], heads[i][1], spec_dim, ap_dim)]
        done = 0
        writes = []
        for fs in sorted({heads[i][1] for i in mine}):
            params = W.default_params(fs, frame_period)
            idx = sorted((i for i in mine if heads[i][1] == fs), key=lambda i: -frames[i])
            for group in _batches(idx, frames, max_batch_frames):
                xs = [x for x, _ in pool.map(lambda i: read_wav(jobs[i][0]), group)]   # this batch only, dropped after it
                b = W.WorldBatch(ctx, params, x_lengths=[len(x) for x in xs])
                x = torch.from_numpy(np.concatenate(xs)).pin_memory().cuda(non_blocking=True)
                del xs
                t, f0, sp, ap = b.analyze(x)
                if spec_dim:
                    f0o, spo, apo = b.recipe_features(f0, sp, ap, spec_dim, ap_dim)         # analysis.cpp:292-366
                else:
                    f0o, spo, apo = f0.float(), sp.float(), ap.float()                      # :360-390
                f0h, sph, aph = (v.cpu().numpy() for v in (f0o, spo, apo))
                fo = b.frame_offsets
                items = []                     # row slices of contiguous slabs: contiguous themselves
                for k, i in enumerate(group):
                    a, e = fo[k], fo[k + 1]
                    items.extend(zip(jobs[i][1:], (f0h[a:e], sph[a:e], aph[a:e])))
                # the batch's files in ONE native call (WorldMi355WriteFiles: plain threads, no interpreter lock),
                # beside the next batch's decoding and analysis; the slabs live on in the job's arguments
                writes.append(pool.submit(W.write_files, items, io_threads))
                done += int(b.total_frames)
                b.close()
        for w_ in writes:
            w_.result()
        if own_ctx:
            ctx.close()
    return done


def _rank_world():
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    return rank, world


def _analysis_gathered(jobs, heads, frame_period, spec_dim, ap_dim, ctx, max_batch_frames, io_threads):
    """configs[3]'s flow for a file list: shard, analyse, gather-v to rank 0, rank 0 writes (needs an initialised
    process group when there is more than one rank)."""
    import torch.distributed as dist
    from . import sweep
    rank, world = _rank_world()
    if world > 1 and not dist.is_initialized():
        raise RuntimeError("gather=True with WORLD_SIZE > 1 needs torch.distributed to be initialised")
    backend = dist.get_backend() if world > 1 else "nccl"
    done = 0
    for fs in sorted({h[1] for h in heads}):
        sel = [i for i, h in enumerate(heads) if h[1] == fs]
        sw = sweep.ShardedSweep(ctx, fs, frame_period, [heads[i][0] for i in sel], rank, world, spec_dim, ap_dim,
                                max_batch_frames, backend)
        sw.load(lambda k: read_wav(jobs[sel[k]][0])[0], io_threads)
        sw.run(sweep.file_sink([jobs[i][1:4] for i in sel]) if rank == 0 else None, io_threads)
        done += sw.my_frames
        sw.close()
    return done


def synth_files(jobs, frame_period, fft_size, fs, spec_dim=0, ap_dim=24, ctx=None,
                max_batch_frames=MAX_BATCH_FRAMES, io_threads=8):
    """jobs: [(f0_in, sp_in, ap_in, wav_out)].  Writes what `synth f0 sp ap wav frame_period fft_size fs
    [spec_dim [ap_dim]]` writes.  In the coded form the ap bins beyond the coding order, which the reference
    leaves uninitialised (synth.cpp:240-245), are 0."""
    import torch
    jobs = list(jobs)
    w = fft_size // 2 + 1
    frames = [os.path.getsize(j[0]) // 4 for j in jobs]                                     # synth.cpp:151-158
    mine = _my_share(frames)
    own_ctx = ctx is None
    ctx = ctx or _own_context()
    params = W.default_params(fs, frame_period, fft_size=fft_size)
    done = 0
    with ThreadPoolExecutor(io_threads) as pool:
        writes = []
        for group in _batches(sorted(mine, key=lambda i: -frames[i]), frames, max_batch_frames):
            T = [frames[i] for i in group]
            ylen = [int((t - 1) * frame_period / 1000.0 * fs) + 1 for t in T]              # synth.cpp:259
            b = W.WorldBatch(ctx, params, f0_lengths=T, y_lengths=ylen)
            cols = (spec_dim, ap_dim) if spec_dim else (w, w)
            f0s, sps, aps = zip(*pool.map(lambda i: (_f32(jobs[i][0]), _f32(jobs[i][1], cols[0]),
                                                     _f32(jobs[i][2], cols[1])), group))
            dev = lambda parts: torch.from_numpy(np.ascontiguousarray(np.concatenate(parts))).cuda()
            if spec_dim:
                f0, sp, ap = b.recipe_decode(dev(f0s), dev(sps), dev(aps))                  # synth.cpp:168-256
            else:
                f0, sp, ap = dev(f0s).double(), dev(sps).double(), dev(aps).double()
            y = b.synthesize(f0, sp, ap).cpu().numpy()
            yo = b.out_offsets
            for k, i in enumerate(group):
                writes.append(pool.submit(write_wav, jobs[i][3], y[yo[k]:yo[k + 1]], fs))
            done += int(b.total_frames)
            b.close()
        for w_ in writes:
            w_.result()
    if own_ctx:
        ctx.close()
    return done


def read_window(path):
    """data/win/NAME.winK: one line, the number of coefficients then the coefficients (window.pl:70-75)."""
    with open(path) as f:
        tok = f.readline().split()
    n = int(tok[0])
    return [float(v) for v in tok[1:1 + n]]


def cmp_files(jobs, streams, sampling_rate, frame_shift, htk_type=9, ctx=None, max_batch_frames=MAX_BATCH_FRAMES,
              io_threads=8):
    """The recipe's `cmp` stage (data/Makefile.in:244-323): window.pl on every stream, the SPTK merge chain, and
    addhtkheader.pl, for a whole list.

    jobs:    [(stream_file_0, ..., stream_file_k, cmp_out)] -- float32 files [T][dim_s] in the order the recipe
             merges them (mgc, lf0, bap, vib); the four must have the same T
    streams: [(dim_s, [window files or coefficient lists])] per stream
    sampling_rate, frame_shift: addhtkheader.pl's SAMPFREQ and FRAMESHIFT (samples)"""
    import torch
    jobs = list(jobs)
    ns = len(streams)
    dims = [int(d) for d, _ in streams]
    wins = [[read_window(w) if isinstance(w, (str, os.PathLike)) else [float(v) for v in w] for w in ws] for _, ws in streams]
    cols = sum(d * len(w) for d, w in zip(dims, wins))
    frames = [os.path.getsize(j[0]) // (4 * dims[0]) for j in jobs]
    mine = _my_share(frames)
    own_ctx = ctx is None
    ctx = ctx or _own_context()
    done = 0
    with ThreadPoolExecutor(io_threads) as pool:
        writes = []
        for group in _batches(sorted(mine, key=lambda i: -frames[i]), frames, max_batch_frames):
            T = [frames[i] for i in group]
            b = W.WorldBatch(ctx, W.default_params(sampling_rate, 5.0), f0_lengths=T)
            dev = []
            for s_ in range(ns):
                parts = list(pool.map(lambda i: _f32(jobs[i][s_], dims[s_]), group))
                for i, a in zip(group, parts):
                    if len(a) != frames[i]:
                        raise ValueError("%s has %d frames, %s has %d" % (jobs[i][s_], len(a), jobs[i][0], frames[i]))
                dev.append((torch.from_numpy(np.ascontiguousarray(np.concatenate(parts))).cuda(), wins[s_]))
            out = b.compose_cmp(dev).cpu().numpy()
            fo = b.frame_offsets

            def put(path, rows):
                with open(path, "wb") as f:
                    f.write(W.htk_header(len(rows), sampling_rate, frame_shift, 4 * cols, htk_type))
                    f.write(np.ascontiguousarray(rows).tobytes())
            for k, i in enumerate(group):
                writes.append(pool.submit(put, jobs[i][ns], out[fo[k]:fo[k + 1]]))
            done += int(b.total_frames)
            b.close()
        for w_ in writes:
            w_.result()
    if own_ctx:
        ctx.close()
    return done


# ---- vibrato (data/scripts/Extract.py, data/Makefile.in:215) --------------------------------------------------------
_SCALE = ("C", "Db", "D", "Eb", "E", "F", "Gb", "G", "Ab", "A", "Bb", "B")


def note_pitch(note):
    """Hz of a note name such as "A4" or "Db5" (equal temperament around A4 = 440; Extract.py:109-114); 0 for "xx"."""
    if note == "xx":
        return 0.0
    return 440.0 * 2.0 ** (int(note[-1:]) - 4) * 2.0 ** ((_SCALE.index(note[:-1]) - 9) / 12.0)


def read_label_segments(mono_path, full_path, frame_period, n_frames):
    """[(start_frame, end_frame, note_pitch_hz)] of an utterance from its mono and full-context label files, as
    Extract.py reads them (:63-84, :176-187): three fields per line, times in units of 100 ns divided by 10e3 (ms),
    frames = floor(time / frame_period) clamped to [0, n_frames], the note from the `/E:<note>]` field of the
    full-context label."""
    import math
    import re
    with open(mono_path) as f:
        mono = f.read().split("\n")
    with open(full_path) as f:
        full = f.read().split("\n")
    if len(mono) != len(full):
        raise ValueError("mono label not equal with full label")
    out = []
    for m, fl in zip(mono, full):
        if m == "" or fl == "":
            continue
        md, fd = m.split(" "), fl.split(" ")
        if len(md) != 3 or len(fd) != 3:
            raise ValueError("label line without three fields")
        t0, t1 = float(md[0]) / 10e3, float(md[1]) / 10e3
        note = re.findall(r"/E:\w+\]", fd[2])[0].replace("/E:", "").replace("]", "")
        out.append((max(math.floor(t0 / frame_period), 0), min(math.floor(t1 / frame_period), n_frames), note_pitch(note)))
    return out


def vibrato_files(jobs, frame_period, fs=48000, ctx=None, max_batch_frames=MAX_BATCH_FRAMES, io_threads=8):
    """jobs: [(lf0_file, mono_label, full_label, vib_out)].  What `Extract.py <base> <frame_period>` does for every
    job: the lf0 file (float32 log f0, one column) is REWRITTEN with two columns (log f0, log(f0 - note + 500)), the
    vib file gets (log depth, log period) per frame.  fs only labels the batch (nothing here depends on it)."""
    import torch
    jobs = list(jobs)
    frames = [os.path.getsize(j[0]) // 4 for j in jobs]
    mine = _my_share(frames)
    own_ctx = ctx is None
    ctx = ctx or _own_context()
    done = 0
    with ThreadPoolExecutor(io_threads) as pool:
        writes = []
        for group in _batches(sorted(mine, key=lambda i: -frames[i]), frames, max_batch_frames):
            T = [frames[i] for i in group]
            b = W.WorldBatch(ctx, W.default_params(fs, frame_period), f0_lengths=T)
            lf0s = list(pool.map(lambda i: _f32(jobs[i][0]), group))
            segs = [read_label_segments(jobs[i][1], jobs[i][2], frame_period, frames[i]) for i in group]
            lf0 = torch.from_numpy(np.ascontiguousarray(np.concatenate(lf0s))).cuda()
            vib, lf2, too_long = b.vibrato(lf0, segs)
            if too_long:
                print("warning: %d voiced run(s) longer than 3072 frames left without vibrato" % too_long, file=sys.stderr)
            vh, lh = vib.cpu().numpy(), lf2.cpu().numpy()
            fo = b.frame_offsets
            for k, i in enumerate(group):
                writes.append(pool.submit(np.ascontiguousarray(lh[fo[k]:fo[k + 1]]).tofile, jobs[i][0]))
                writes.append(pool.submit(np.ascontiguousarray(vh[fo[k]:fo[k + 1]]).tofile, jobs[i][3]))
            done += int(b.total_frames)
            b.close()
        for w_ in writes:
            w_.result()
    if own_ctx:
        ctx.close()
    return done


def _read_scp(path):
    with open(path) as f:
        rows = [ln.split() for ln in f if ln.strip() and not ln.startswith("#")]
    bad = [r for r in rows if len(r) != 4]
    if bad:
        raise SystemExit("every line needs four paths: %r" % (bad[0],))
    return [tuple(r) for r in rows]


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m hts-train-world_amd.recipe", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("analysis", "synth"):
        p = sub.add_parser(name)
        p.add_argument("--scp", required=True, help="job list, four paths per line in the CLI's argument order")
        p.add_argument("--frame-period", type=float, default=5.0)
        p.add_argument("--fft-size", type=int, default=0 if name == "analysis" else None, required=name == "synth")
        p.add_argument("--spec-dim", type=int, default=0, help="0: uncompressed files")
        p.add_argument("--ap-dim", type=int, default=24)
        if name == "analysis":
            p.add_argument("--gather", action="store_true",
                           help="several ranks (torchrun): gather the features to rank 0, which writes every file")
            p.add_argument("--resume", action="store_true",
                           help="skip utterances whose three output files are already complete (a run that was cut short)")
        if name == "synth":
            p.add_argument("--fs", type=int, required=True)
    a = ap.parse_args(argv)
    jobs = _read_scp(a.scp)
    if a.cmd == "analysis":
        if a.gather and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
            dist.init_process_group(os.environ.get("WM_BACKEND", "nccl"))
        n = analysis_files(jobs, a.frame_period, a.fft_size, a.spec_dim, a.ap_dim, gather=a.gather, resume=a.resume)
    else:
        n = synth_files(jobs, a.frame_period, a.fft_size, a.fs, a.spec_dim, a.ap_dim)
    print("complete. %d frames" % n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
