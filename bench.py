#!/usr/bin/env python3
"""bench.py -- WORLD analysis+synthesis frames/sec on MI355X (BASELINE.json's metric).

One "step" = one pass of the hot path over one batch of synthetic 16 kHz utterances that is
already resident in HBM: Dio -> StoneMask -> CheapTrick -> D4C (analysis) then Synthesis from the
features just produced, every utterance of the batch, fp64 as the reference computes.  Workload at
N=1 is BASELINE.json configs[1]: 256 synthetic utterances of 2-8 s.  With --gpus N every rank owns
its own 256 utterances (weak scaling, no data-path collective); `--gather` adds the RCCL gather-v
of the float32 feature files to rank 0 inside the step.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     -- dominant kernel (d4c_kernel): algorithmic bytes per launch / HIP-event duration
                  measured on the launch stream, against the 8 TB/s HBM peak; the FP64 figures that
                  actually bound the path are reported beside it.
  cpu_baseline -- the reference WORLD (oracle/_ref, kind "reference") or this repo's C restatement
                  (kind "port") timed single-threaded on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector peak
FLOPS_PER_FRAME = 0.7e6        # SURVEY.md section 8d (16 kHz)


def byte_model(fs, fp_ms, fft_size):
    """Algorithmic HBM bytes per frame, fp64 C-API layout:
      analysis  : hop samples*8 + t 8 + f0 8 + sp bins*8 + ap bins*8   (8864 at 16 kHz / 5 ms / fft 1024)
      synthesis : f0 8 + sp + ap + hop samples*8                       (8856)
      d4c_kernel: its hop new samples + t + f0 + ap0 in, one ap row out (4768)"""
    hop = fs * fp_ms / 1000.0
    bins = fft_size // 2 + 1
    analysis = hop * 8 + 16 + 2 * bins * 8
    synthesis = 8 + 2 * bins * 8 + hop * 8
    d4c = hop * 8 + 24 + bins * 8
    return {"analysis": analysis, "synthesis": synthesis, "round_trip": analysis + synthesis, "d4c": d4c}


def d4c_flops_per_voiced_frame(fs):
    """Real FFTs of the D4C body per voiced frame: 2 x 2 (centroids) + 1 (power) + one per band, each
    2.5 N log2 N (scans, windows and the rest are not counted)."""
    fd = 2 ** (1 + int(np.log2(4.0 * fs / 47.0 + 1)))                       # d4c.cpp:344-346
    bands = int(min(15000.0, fs / 2.0 - 3000.0) / 3000.0)                  # d4c.cpp:351-353
    return (5 + bands) * 2.5 * fd * np.log2(fd)


# HBM-side bytes of the dominant kernel come from PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
# runs, tools/pmc_hbm.sh) stored in profiles/pmc_traffic.json together with a hash of the kernel's sources: when the
# sources have changed since the counters were collected the figure is stale and `traffic` is reported as null.
D4C_SOURCES = ("d4c.hip", "d4c_big.hpp", "peel.hpp", "fft.hpp", "frame.hpp", "spectrum.hpp", "common.hpp", "window.hpp",
               "partition.hpp", "wavesync.hpp")


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for n in D4C_SOURCES:
        with open(os.path.join(ROOT, "hts-train-world_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(fs):
    """(bytes per frame, note) from profiles/pmc_traffic.json, or (None, reason)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            recs = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/pmc_traffic.json missing"
    sha = kernel_source_hash()
    mine = [r for r in recs if r.get("fs") == fs]
    if not mine:
        return None, "no PMC passes recorded for fs %d in profiles/pmc_traffic.json" % fs
    for r in mine:
        if r.get("source_sha") == sha:
            per = (r["fetch_kb"] + r["write_kb"]) * 1024.0 / r["frames"]
            return per, "PMC FETCH_SIZE + WRITE_SIZE of %s, separate passes (%s), %d frames" % (
                r["kernel"], r.get("files", "profiles/"), r["frames"])
    return None, "kernel sources changed since the PMC passes in profiles/pmc_traffic.json (now %s)" % sha


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks, one per GPU.  Under torchrun (WORLD_SIZE set) this is informational; started plainly with "
                         "--gpus N > 1 the script itself starts N rank processes before anything touches the GPU")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prewarm", type=float, default=10.0,
                    help="seconds of the same step run untimed before the warm-up steps (0 = none): an MI355X takes "
                         "about ten seconds of load to reach its sustained clock (tools/step_trace.py: 16.9 ms per step, "
                         "then 16.0); the first steps are timed as well and reported as clock_ramp.cold_ms_per_step")
    ap.add_argument("--utts", type=int, default=0, help="utterances per GPU (sweep: of the whole corpus); 0 = the workload's own")
    ap.add_argument("--dur", type=float, nargs=2, default=(2.0, 8.0))
    ap.add_argument("--fs", type=int, default=16000, choices=(16000, 48000),
                    help="analysis_synthesis only: 16000 is BASELINE.json's metric; 48000 (fft 2048, D4C fft 4096) is the "
                         "setting the reference's author committed (config.status: SAMPFREQ=48000)")
    ap.add_argument("--gather", action="store_true", help="include the RCCL feature gather in the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate-calls", action="store_true",
                    help="WorldMi355Analyze then WorldMi355Synthesis instead of WorldMi355AnalyzeSynthesize")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl (= RCCL; the driver's launch) or gloo: rehearsal of the N > 1 control path with several "
                         "ranks sharing one GPU, where RCCL refuses duplicate devices")
    ap.add_argument("--cpu-utts", type=int, default=48, help="utterances of the CPU baseline sample")
    ap.add_argument("--workers", type=int, default=0, help="processes for synthetic data generation (0 = auto)")
    ap.add_argument("--workload", choices=["analysis_synthesis", "sweep", "harvest", "synthesis", "codec"],
                    default="analysis_synthesis",
                    help="analysis_synthesis = configs[1] (the headline metric); sweep = configs[3]: a fixed corpus of ~1000 "
                         "utterances sharded over the ranks (LPT), analysed, gathered to rank 0 as float32 and written to files "
                         "by rank 0 (strong scaling); harvest = configs[2] (48 kHz, 1 ms, 64 utterances); synthesis = configs[4] "
                         "(Synthesis only from precomputed features); codec = the recipe's coded lf0/mgc/bap from resident "
                         "features plus the decoders (SURVEY.md 8(f))")
    ap.add_argument("--coded", action="store_true", help="sweep: write the recipe's coded lf0/mgc/bap (50 + 25 dims) instead of raw f0/sp/ap")
    ap.add_argument("--rounds", type=int, default=4,
                    help="sweep: batches per rank; rank 0 copies and writes one round while the next is analysed")
    ap.add_argument("--writers", choices=("rank0", "all"), default="rank0",
                    help="sweep: rank0 = configs[3] as stated (gather-v, rank 0 writes everything); all = no gather, every "
                         "rank writes its own shard's files (shows what the rank-0 funnel costs)")
    ap.add_argument("--plan-only", action="store_true",
                    help="print this rank's share of the workload as JSON and exit without touching the GPU")
    ap.add_argument("--out-dir", default=None, help="sweep: where rank 0 writes the feature files (default: a fresh temp dir, removed afterwards)")
    args = ap.parse_args()
    if args.utts <= 0:
        args.utts = {"sweep": 1000, "harvest": 64, "synthesis": 1024}.get(args.workload, 256)
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start N rank processes of this script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, rank r on GPU r) and wait for them.  Runs before this process has
    made any HIP call (torch.cuda.device_count() does not initialise the runtime), and the ranks are fresh
    children, never an exec of a process that holds the GPU."""
    import socket
    import subprocess
    import torch
    n = args.gpus
    have = torch.cuda.device_count()
    if args.backend == "nccl" and have < n:
        sys.exit("bench.py: --gpus %d but %d GPU(s) visible; RCCL needs one device per rank "
                 "(--backend gloo rehearses the control path with ranks sharing a GPU)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    sys.exit(rc)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    pkg = importlib.import_module("hts-train-world_amd")
    W, sd, sh = pkg.world, pkg.synth_data, pkg.sharding

    # ---- synthetic workload (not timed).  Generated BEFORE the GPU is touched: the generator forks a
    # worker pool, and a process that has initialised HIP / RCCL should not be forked ----
    fs, fp, utts = workload_spec(args)
    ncpu = os.cpu_count() or 1
    workers = args.workers if args.workers > 0 else max(1, min(16, ncpu // max(1, world)))
    if args.workers <= 0 and any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES")):
        workers = 1      # under rocprofv3 the preloaded tool has already initialised the GPU: do not fork
    plan = None
    if args.workload == "sweep":
        # a FIXED corpus whatever the number of ranks (strong scaling): every rank derives the same partition from
        # the lengths alone and generates only its own utterances
        counts = [sd.utterance_samples(i, fs, tuple(args.dur)) for i in range(utts)]
        frames_all = [sh.frame_count(n, fs, fp) for n in counts]
        mine = sh.lpt_shards(frames_all, world)[rank]
        plan = (counts, mine)
        if args.plan_only:
            print(json.dumps({"rank": rank, "world": world, "local_rank": local, "workload": "sweep", "utterances": mine,
                              "frames": sum(frames_all[i] for i in mine), "corpus_frames": sum(frames_all)}), flush=True)
            return
        xs = sd.make_batch(len(mine), fs, tuple(args.dur), workers=workers, indices=mine)
    else:
        if args.plan_only:
            print(json.dumps({"rank": rank, "world": world, "local_rank": local, "workload": args.workload,
                              "utterances": list(range(rank * utts, (rank + 1) * utts))}), flush=True)
            return
        xs = sd.make_batch(utts, fs, tuple(args.dur), first=rank * utts, workers=workers)
    # the CPU path on all host cores of this process' share, also before the GPU is touched
    cpu_all = None
    if (world == 1 and workers > 1 and not args.no_cpu_baseline and args.workload == "analysis_synthesis"):
        cpu_all = cpu_all_cores(xs, fs, fp, workers, 6 * workers)

    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    shared_gpu = local >= torch.cuda.device_count()
    if args.backend == "gloo":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    if args.workload == "sweep":
        return sweep_workload(args, torch, dist, pkg, rank, world, xs, plan, fs, fp, shared_gpu)
    if args.workload != "analysis_synthesis":
        return side_workload(args, torch, dist, W, sd, rank, world, xs, fs, fp, utts)

    lens = [len(x) for x in xs]
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
    batch = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=lens)
    frames = int(batch.total_frames)
    outs = (torch.empty(frames, dtype=torch.float64, device="cuda"), torch.empty(frames, dtype=torch.float64, device="cuda"),
            torch.empty(frames, batch.bins, dtype=torch.float64, device="cuda"),
            torch.empty(frames, batch.bins, dtype=torch.float64, device="cuda"))
    y = torch.empty(int(batch.total_out), dtype=torch.float64, device="cuda")
    frame_counts = np.diff(batch.frame_offsets).tolist()
    bm = byte_model(fs, fp, batch.fft_size)

    def compute():
        if args.separate_calls:
            t, f0, sp, ap = batch.analyze(x, out=outs)
            batch.synthesize(f0, sp, ap, out=y)
        else:
            # the same launches as one call: Synthesis' f0-only first part runs beside CheapTrick / D4C
            t, f0, sp, ap, _ = batch.analyze_synthesize(x, out=outs, y=y)
        return f0, sp, ap

    def step():
        f0, sp, ap = compute()
        if args.gather and world > 1:
            # the on-disk types of the reference CLI are float32 (test/analysis.cpp:360-390)
            feats = [f0.float(), sp.float(), ap.float()]
            if args.backend == "gloo":
                feats = [v.cpu() for v in feats]
            sh.gather_features(feats, frame_counts, dst=0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the ramp runs the rank-local part only: its length is set by each rank's clock, so it must not hold a collective
    ramp = clock_ramp(compute, barrier, args.prewarm, args.steps) if args.prewarm > 0 else None
    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = {k: ctx.timing_query(k) for k in ANALYSIS_KERNELS + SYNTHESIS_KERNELS}
    ctx.timing_enable(False)
    elapsed_max, total_frames = reduce_over_ranks(torch, dist, world, args.backend, elapsed, frames)
    hi = host_inclusive(args, torch, dist, pkg, ctx, xs, fs, fp, world)
    hic = host_inclusive(args, torch, dist, pkg, ctx, xs, fs, fp, world, coded=(50, 25))

    if rank == 0:
        value = total_frames * args.steps / elapsed_max
        voiced = int((outs[1] > 0).sum().item())
        roof = d4c_roofline(kernel_ms, frames, voiced, fs, bm, args.steps)
        roof["pipeline"] = {"achieved_gbs": round(value / world * bm["round_trip"] / 1e9, 3),
                            "frac_hbm": round(value / world * bm["round_trip"] / 1e9 / HBM_PEAK_GBS, 6),
                            "achieved_fp64_tflops": round(value / world * FLOPS_PER_FRAME / 1e12, 4) if fs == 16000 else None}
        cpu = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, parity = cpu_baseline_and_parity(xs, fs, fp, batch, outs, y, args.cpu_utts)
        line = {
            "metric": "WORLD analysis+synthesis frames/sec @%dkHz, 5ms hop" % (fs // 1000),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: batch of %d synthetic %d kHz utterances (%g-%g s) per GPU, "
                                   "Dio+StoneMask+CheapTrick+D4C then Synthesis (%s), fp64, features resident in HBM"
                                   % (args.utts, fs // 1000, args.dur[0], args.dur[1],
                                      "two calls" if args.separate_calls else
                                      "one call: Synthesis' f0-only part on a second stream beside CheapTrick/D4C"),
                       "fs": fs, "frame_period_ms": fp, "fft_size": batch.fft_size,
                       "utterances_per_gpu": args.utts, "frames_per_gpu": frames,
                       "parallelism": "utterance-sharded x%d%s" % (world, "+gather" if args.gather else "")},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if ramp:
            line["clock_ramp"] = ramp
        if shared_gpu or (world > 1 and args.backend == "gloo"):
            line["config"]["note"] = "rehearsal: %d ranks over gloo on %d GPU(s)" % (world, torch.cuda.device_count())
        line["host_inclusive"] = hi
        line["host_inclusive_coded"] = hic
        if cpu_all:
            line["cpu_baseline_all_cores"] = cpu_all
        if parity:
            line["parity"] = parity
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def host_inclusive(args, torch, dist, pkg, ctx, xs, fs, fp, world, coded=None):
    """The same step fed from and drained to pinned HOST memory in the on-disk types (int16 samples up; float32
    f0 / sp / ap and int16 resynthesised samples down), double-buffered on copy streams beside the kernels
    (hts-train-world_amd/pipeline.py).  Reported beside `value`, never as it (the metric
    "including H2D of waveforms and D2H of features")."""
    pl = pkg.pipeline
    pipe = pl.HostPipeline(ctx, pkg.world.default_params(fs, fp), [len(x) for x in xs], synthesis=True, coded=coded)
    x16 = pl.to_int16(np.concatenate(xs))
    for xb in pipe.x_pinned:                        # the waveforms wait in pinned memory, as decoded wav payloads would
        xb.numpy()[:] = x16
    steps = max(2, args.steps)
    for _ in range(2):
        pipe.result(pipe.submit())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prev = None
    pipe.feed()
    for k in range(steps):
        if k + 1 < steps:
            pipe.feed()                             # the next step's upload goes ahead of this step's download
        slot = pipe.submit()
        if prev is not None:
            pipe.result(prev)
        prev = slot
    pipe.result(prev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = int(pipe.batch.total_frames)
    up, down = pipe.bytes_per_step()
    pipe.close()
    dt_max, total = reduce_over_ranks(torch, dist, world, args.backend, dt, frames)
    return {"value": round(total * steps / dt_max, 1), "unit": "frames/s", "ms_per_step": round(dt_max / steps * 1e3, 3),
            "overlapped": True, "steps": steps,
            "layout": "pinned host int16 waveforms up (%d B/frame), float32 %s + int16 resynthesis down (%d B/frame); "
                      "two slots, uploads / kernels / downloads on three streams"
                      % (up // max(1, frames), "lf0/mgc[%d]/bap[%d] (the recipe's coded features)" % coded if coded else "f0/sp/ap",
                         down // max(1, frames)),
            "pcie_gbs": round((up + down) * steps / dt / 1e9, 2)}


def clock_ramp(step, barrier, seconds, steps):
    """Time the first `steps` steps as they come (after one untimed step that takes the first-call allocations),
    then keep stepping untimed until `seconds` have passed.  Returns the `clock_ramp` object of the JSON line."""
    step()
    barrier()
    t0 = time.perf_counter()
    n = max(1, min(steps, 5))
    for _ in range(n):
        step()
    barrier()
    cold = (time.perf_counter() - t0) / n
    import torch
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < seconds:      # per rank by the clock: device synchronisation only, no collective
        for _ in range(8):
            step()
        torch.cuda.synchronize()
    barrier()
    return {"prewarm_s": seconds, "cold_ms_per_step": round(cold * 1e3, 3),
            "note": "value is measured after prewarm_s seconds of the same step (sustained clock); cold_ms_per_step is "
                    "the first %d steps of the process" % n}


ANALYSIS_KERNELS = ("dio_lowcut_kernel", "dio_band_kernel", "dio_candidate_kernel", "dio_fix_kernel", "stonemask_kernel",
                    "cheaptrick_kernel", "d4c_lovetrain_kernel", "d4c_kernel")
SYNTHESIS_KERNELS = ("synth_inc_kernel", "synth_timebase_kernel", "synth_search_kernel", "synth_pulse_kernel",
                     "synth_ola_kernel")


def reduce_over_ranks(torch, dist, world, backend, elapsed, frames):
    """(max elapsed over ranks, total frames over ranks)."""
    if world == 1:
        return elapsed, float(frames)
    tt = torch.tensor([elapsed, float(frames)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    tmax, tsum = tt.clone(), tt.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return float(tmax[0]), float(tsum[1])


def d4c_roofline(kernel_ms, frames, voiced, fs, bm, steps):
    """The `roofline` object for the dominant kernel (d4c_kernel): algorithmic bytes per launch over the HIP-event
    duration of its launches on the launch stream, against the HBM peak; FP64 figures (what binds it) beside it."""
    d4c_ms, d4c_n = kernel_ms["d4c_kernel"]
    d4c_avg_s = d4c_ms / max(1, d4c_n) * 1e-3
    # a step may launch the kernel several times (the sweep analyses its shard in batches): per-launch units
    per_step = max(1, round(d4c_n / max(1, steps)))
    frames = frames / per_step
    voiced = voiced / per_step
    per_frame, note = measured_traffic(fs)
    flops = d4c_flops_per_voiced_frame(fs)
    ok = d4c_avg_s > 0
    fd = 2 ** (1 + int(math.log2(4.0 * fs / 47.0 + 1.0)))          # fft_size_d4c (d4c.cpp:341-343)
    name = "d4c_kernel" if fd <= 2048 else "d4c_kernel scope = d4cb_centroid + d4cb_spectrum + d4cb_band + d4cb_output (fft %d)" % fd
    return {
        "bound": "hbm", "kernel": name,
        "achieved": round(frames * bm["d4c"] / d4c_avg_s / 1e9, 3) if ok else None,
        "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(frames * bm["d4c"] / d4c_avg_s / 1e9 / HBM_PEAK_GBS, 6) if ok else None,
        "traffic": round(frames * per_frame) if per_frame is not None else None,
        "launches_per_step": per_step,
        "traffic_note": note,
        "launch_ms": round(d4c_avg_s * 1e3, 4), "units_per_launch": round(frames),
        "bytes_per_unit": bm["d4c"],
        "note": "FP64-FFT/LDS bound, not HBM bound (SURVEY.md 8d); fp64 figures beside it",
        "fp64": {"achieved_tflops": round(voiced * flops / d4c_avg_s / 1e12, 3) if ok else None,
                 "peak_tflops": FP64_PEAK_TFLOPS,
                 "frac": round(voiced * flops / d4c_avg_s / 1e12 / FP64_PEAK_TFLOPS, 5) if ok else None},
        "kernel_ms_per_step": {k: round(v[0] / steps, 4) for k, v in kernel_ms.items()},
    }


def sweep_workload(args, torch, dist, pkg, rank, world, xs, plan, fs, fp, shared_gpu):
    """configs[3]: the data/ feature-extraction sweep over a fixed corpus (data/Makefile.in:125-242), sharded over the
    ranks; one step = every rank analyses its shard, the float32 feature slabs are gathered to rank 0, rank 0 writes
    every utterance's files.  Waveforms are resident in HBM when the timed region starts."""
    import shutil
    import tempfile
    W, sweep = pkg.world, pkg.sweep
    counts, mine = plan
    ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
    sw = sweep.ShardedSweep(ctx, fs, fp, counts, rank, world, spec_dim=50 if args.coded else 0, ap_dim=25,
                            backend=args.backend, rounds=args.rounds, writers=args.writers)
    by_id = dict(zip(mine, xs))
    sw.load(lambda i: by_id[i])
    out_dir = None
    sink = None
    if rank == 0:
        out_dir = args.out_dir or tempfile.mkdtemp(prefix="wm_sweep_")
    if args.writers == "all" and world > 1:
        box = [out_dir]
        dist.broadcast_object_list(box, src=0)
        out_dir = box[0]
    if out_dir:
        names = ("lf0", "mgc", "bap") if args.coded else ("f0", "sp", "ap")
        sink = sweep.dir_sink(out_dir, names)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sw.run(sink)
    barrier()
    ctx.timing_enable(True)
    phases = {"compute": 0.0, "gather": 0.0, "to_host": 0.0, "write": 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ph = sw.run(sink)
        for k in phases:
            phases[k] += ph[k]
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = {k: ctx.timing_query(k) for k in ANALYSIS_KERNELS}
    ctx.timing_enable(False)
    elapsed_max, _ = reduce_over_ranks(torch, dist, world, args.backend, elapsed, sw.my_frames)
    # slowest rank's compute, the busiest rank's share of frames (LPT balance)
    comp_max, _ = reduce_over_ranks(torch, dist, world, args.backend, phases["compute"], 0)
    most, _ = reduce_over_ranks(torch, dist, world, args.backend, float(sw.my_frames), 0)
    if rank == 0:
        total = sw.total_frames
        value = total * args.steps / elapsed_max
        F = pkg.capi.cheaptrick_fft_size(fs)
        bm = byte_model(fs, fp, F)
        frames0 = sum(int(b.total_frames) for b, _ in sw.loaded)
        voiced = 0
        for b, xx in sw.loaded:
            voiced += int((b.analyze(xx)[1] > 0).sum().item())
        roof = d4c_roofline(kernel_ms, frames0, voiced, fs, bm, args.steps) if frames0 else None
        per_frame_out = 4 * ((1 + 50 + 25) if args.coded else (1 + 2 * (F // 2 + 1)))
        k = float(args.steps)
        line = {
            "metric": "WORLD analysis sweep frames/sec @16kHz, 5ms hop (corpus -> float32 feature files on rank 0)",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[3]: fixed corpus of %d synthetic 16 kHz utterances (%g-%g s, %d frames), LPT-sharded "
                                   "over %d rank(s); Dio+StoneMask+CheapTrick+D4C in fp64, %s float32 gathered to rank 0 "
                                   "(grouped send/recv), written by rank 0 as one file per utterance and feature"
                                   % (len(counts), args.dur[0], args.dur[1], total, world,
                                      "coded lf0/mgc(50)/bap(25)" if args.coded else "raw f0/sp/ap"),
                       "fs": fs, "frame_period_ms": fp, "utterances": len(counts), "frames": total,
                       "frames_on_busiest_rank": int(most), "output_bytes_per_frame": per_frame_out,
                       "rounds": sw.rounds, "writers": args.writers,
                       "parallelism": "utterance-sharded x%d, %s" % (world, "gather-v to rank 0 over " + args.backend
                                                                      if args.writers == "rank0" else "every rank writes its shard")},
            "phases_ms_per_step": {"compute_slowest_rank": round(comp_max / k * 1e3, 3),
                                   "gather_rank0": round(phases["gather"] / k * 1e3, 3),
                                   "to_host_rank0": round(phases["to_host"] / k * 1e3, 3),
                                   "file_write_rank0": round(phases["write"] / k * 1e3, 3)},
            "value_compute_only": round(total * k / comp_max, 1) if comp_max > 0 else None,
            "roofline": roof, "cpu_baseline": None,
        }
        if shared_gpu or (world > 1 and args.backend == "gloo"):
            line["config"]["note"] = "rehearsal: %d ranks over gloo on %d GPU(s)" % (world, torch.cuda.device_count())
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["parity"] = sweep_cpu_baseline(by_id, mine, fs, fp, out_dir, args)
        print(json.dumps(line), flush=True)
        if not args.out_dir:
            shutil.rmtree(out_dir, ignore_errors=True)
    sw.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def sweep_cpu_baseline(by_id, mine, fs, fp, out_dir, args):
    """The reference's `analysis` arithmetic (Dio+StoneMask+CheapTrick+D4C, float32 outputs) single-threaded on a
    bounded sample of the corpus, and the files rank 0 wrote for those utterances checked against it."""
    from oracle.bindings import Oracle, Reference
    lib = Reference() if Reference.available() else Oracle()
    F = lib.cheaptrick_fft_size(fs)
    frames, tcpu, df0, se_sp, se_ap, cnt, n = 0, 0.0, 0.0, 0.0, 0.0, 0, 0
    for i in mine[:args.cpu_utts]:
        x = by_id[i]
        a = time.perf_counter()
        t, f0 = lib.dio(x, fs, fp)
        f0 = lib.stonemask(x, fs, t, f0)
        sp = lib.cheaptrick(x, fs, t, f0, -0.15, F)
        ap = lib.d4c(x, fs, t, f0, F, 0.0)
        tcpu += time.perf_counter() - a
        frames += len(f0)
        n += 1
        if not args.coded:
            g_f0 = np.fromfile(os.path.join(out_dir, "utt%05d.f0" % i), dtype=np.float32)
            g_sp = np.fromfile(os.path.join(out_dir, "utt%05d.sp" % i), dtype=np.float32).reshape(-1, F // 2 + 1)
            g_ap = np.fromfile(os.path.join(out_dir, "utt%05d.ap" % i), dtype=np.float32).reshape(-1, F // 2 + 1)
            df0 = max(df0, float(np.abs(g_f0 - f0.astype(np.float32)).max()))
            se_sp += float(((g_sp.astype(np.float64) - sp.astype(np.float32)) ** 2).sum())
            se_ap += float(((g_ap.astype(np.float64) - ap.astype(np.float32)) ** 2).sum())
            cnt += sp.size
        if tcpu > 25.0:
            break
    cpu = {"value": round(frames / tcpu, 1), "unit": "frames/s", "cores": 1, "kind": lib.kind,
           "sample": "%d utterances of the corpus (%d frames), Dio+StoneMask+CheapTrick+D4C, single thread, %.1f s "
                     "(file writing not included)" % (n, frames, tcpu)}
    parity = None
    if cnt:
        parity = {"vs": lib.kind, "utterances": n, "files": "float32 as written by rank 0",
                  "max_abs_dF0_hz": df0, "sp_rmse": (se_sp / cnt) ** 0.5, "ap_rmse": (se_ap / cnt) ** 0.5}
    return cpu, parity


def workload_spec(args):
    """(fs, frame period, utterances per GPU -- for the sweep: of the corpus) of the selected workload."""
    if args.workload == "harvest":
        return 48000, 1.0, args.utts
    if args.workload == "analysis_synthesis":
        return args.fs, 5.0, args.utts
    return 16000, 5.0, args.utts


def side_workload(args, torch, dist, W, sd, rank, world, xs, fs, fp, utts):
    """configs[2] (Harvest), configs[4] (Synthesis only) and the feature codec: same contract, separate metric names."""
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
    batch = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(v) for v in xs])
    frames = int(batch.total_frames)
    if args.workload == "harvest":
        names = ("hv_decimate", "hv_band_kernel", "hv_raw_kernel", "hv_refine_kernel", "hv_contour_kernel")
        step = lambda: batch.harvest(x)
    elif args.workload == "codec":
        # the recipe's coded features from resident sp/ap, and the decoders back
        t, f0, sp, ap = batch.analyze(x)
        names = ("codec_code_sp_kernel", "codec_decode_sp_kernel", "codec_code_ap_kernel", "codec_decode_ap_kernel")

        def step():
            lf0, mgc, bap = batch.recipe_features(f0, sp, ap, 50, 25)
            csp = batch.code_spectral_envelope(sp, 50)
            cap = batch.code_aperiodicity(ap)
            batch.decode_spectral_envelope(csp)
            batch.decode_aperiodicity(cap)
    else:
        t, f0, sp, ap = batch.analyze(x)
        y = torch.empty(int(batch.total_out), dtype=torch.float64, device="cuda")
        names = ("synth_inc_kernel", "synth_timebase_kernel", "synth_search_kernel", "synth_pulse_kernel", "synth_ola_kernel")
        step = lambda: batch.synthesize(f0, sp, ap, out=y)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ramp = clock_ramp(step, barrier, args.prewarm, args.steps) if args.prewarm > 0 else None
    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kms = {k: ctx.timing_query(k)[0] / args.steps for k in names}
    if args.workload == "codec":
        # the synth CLI's way back (WorldMi355RecipeDecode), outside the timed step: reported beside it
        lf0, mgc, bap = batch.recipe_features(f0, sp, ap, 50, 25)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            batch.recipe_decode(lf0, mgc, bap)
        torch.cuda.synchronize()
        kms["recipe_decode_call_ms (not in value)"] = (time.perf_counter() - t1) * 1e3 / args.steps
        kms["codec_bap_decode_kernel (not in value)"] = ctx.timing_query("codec_bap_decode_kernel")[0] / args.steps
    tt = torch.tensor([elapsed, float(frames)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax, tsum = tt.clone(), tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total = float(tmax[0]), float(tsum[1])
    else:
        total = float(frames)
    if rank == 0:
        value = total * args.steps / elapsed
        bpf = 400 if args.workload == "harvest" else 8856
        if args.workload == "codec":
            # per frame: recipe packing reads sp+ap (2 x 4104 B), coders read them again, decoders write them
            bpf = 4 * 4104 + 2 * 4104
        mname = {"harvest": "Harvest @48kHz, 1ms hop", "codec": "feature codec (recipe packing + code + decode) @16kHz",
                 "synthesis": "Synthesis-only @16kHz, 5ms hop"}[args.workload]
        line = {"metric": "WORLD %s frames/sec" % mname,
                "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "configs[%d]: %d synthetic utterances (%g-%g s) per GPU, fs %d, hop %g ms"
                                       % ({"harvest": 2, "codec": 1}.get(args.workload, 4), utts, args.dur[0], args.dur[1], fs, fp),
                           "frames_per_gpu": frames},
                "roofline": {"bound": "hbm", "achieved": round(value / world * bpf / 1e9, 3), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(value / world * bpf / 1e9 / HBM_PEAK_GBS, 7),
                             "traffic": None, "bytes_per_unit": bpf, "kernel_ms_per_step": {k: round(v, 4) for k, v in kms.items()}},
                "cpu_baseline": None}
        if ramp:
            line["clock_ramp"] = ramp
        if world == 1 and not args.no_cpu_baseline and args.workload == "harvest":
            from oracle.bindings import Oracle, Reference
            lib = Reference() if Reference.available() else Oracle()
            got = batch.harvest(x)[1].cpu().numpy()
            nfr, tcpu, df0 = 0, 0.0, 0.0
            for u in range(min(len(xs), 6)):
                a = time.perf_counter()
                tc, fc = lib.harvest(xs[u], fs, fp)
                tcpu += time.perf_counter() - a
                nfr += len(fc)
                df0 = max(df0, float(np.abs(got[batch.frame_offsets[u]:batch.frame_offsets[u + 1]] - fc).max()))
                if tcpu > 30:
                    break
            line["cpu_baseline"] = {"value": round(nfr / tcpu, 1), "unit": "frames/s", "cores": 1, "kind": lib.kind,
                                    "sample": "first utterances of the same batch (%d frames), %.1f s" % (nfr, tcpu)}
            line["parity"] = {"vs": lib.kind, "max_abs_dF0_hz": df0}
        if world == 1 and not args.no_cpu_baseline and args.workload == "codec":
            from oracle.bindings import Oracle, Reference
            lib = Reference() if Reference.available() else Oracle()
            F = batch.fft_size
            fo = batch.frame_offsets
            n_u = min(len(xs), 24)
            sp_h, ap_h = sp[:fo[n_u]].cpu().numpy(), ap[:fo[n_u]].cpu().numpy()
            got = batch.code_spectral_envelope(sp, 50)[:fo[n_u]].cpu().numpy()
            a = time.perf_counter()
            sp4 = sp_h * 1e4
            sp4[sp4 == 0.0] = 0.0001
            lib.code_spectral_envelope(sp4, fs, F, 50)
            lib.code_spectral_envelope(ap_h * 1e4, fs, F, 25)
            csp = lib.code_spectral_envelope(sp_h, fs, F, 50)
            cap = lib.code_aperiodicity(ap_h, fs, F)
            lib.decode_spectral_envelope(csp, fs, F)
            lib.decode_aperiodicity(cap, fs, F)
            tcpu = time.perf_counter() - a
            line["cpu_baseline"] = {"value": round(fo[n_u] / tcpu, 1), "unit": "frames/s", "cores": 1, "kind": lib.kind,
                                    "sample": "the same step on the first %d utterances (%d frames), %.1f s"
                                              % (n_u, fo[n_u], tcpu)}
            line["parity"] = {"vs": lib.kind, "max_abs_d_coded_sp": float(np.abs(got - csp).max())}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_CPU_XS = None


def _cpu_chain(job):
    """Worker of cpu_all_cores(): the CPU chain on a slice of the parent's utterances (forked: shared pages)."""
    lo, hi, fs, fp = job
    from oracle.bindings import Oracle, Reference
    lib = Reference() if Reference.available() else Oracle()
    F = lib.cheaptrick_fft_size(fs)
    frames = 0
    t0 = time.perf_counter()
    for x in _CPU_XS[lo:hi]:
        t, f0 = lib.dio(x, fs, fp)
        f0 = lib.stonemask(x, fs, t, f0)
        sp = lib.cheaptrick(x, fs, t, f0, -0.15, F)
        ap = lib.d4c(x, fs, t, f0, F, 0.0)
        lib.synthesis(f0, sp, ap, F, fp, fs)
        frames += len(f0)
    return frames, time.perf_counter() - t0, lib.kind


def cpu_all_cores(xs, fs, fp, cores, n_utts):
    """The same CPU chain with one worker process per host core of this box's share, each on its own slice of
    the first n_utts utterances (the reference is single-threaded: one process per utterance is how the recipe
    itself would use more cores).  Reported beside the single-thread baseline, not instead of it."""
    global _CPU_XS
    import multiprocessing as mp
    n_utts = min(n_utts, len(xs))
    per = max(1, n_utts // cores)
    jobs = [(k * per, min(n_utts, (k + 1) * per), fs, fp) for k in range(cores) if k * per < n_utts]
    _CPU_XS = xs
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        res = pool.map(_cpu_chain, jobs)
    wall = time.perf_counter() - t0
    _CPU_XS = None
    frames = sum(r[0] for r in res)
    slowest = max(r[1] for r in res)
    return {"value": round(frames / slowest, 1), "unit": "frames/s", "cores": len(jobs), "kind": res[0][2],
            "sample": "%d utterances of the same batch (%d frames) over %d worker processes, slowest worker %.1f s "
                      "(%.1f s wall with start-up)" % (jobs[-1][1], frames, len(jobs), slowest, wall)}


def cpu_baseline_and_parity(xs, fs, fp, batch, outs, y, n_cpu):
    """Time the CPU path single-threaded on the first n_cpu utterances of the same workload and
    check the GPU results of those utterances against it (the checker, never the thing shipped)."""
    from oracle.bindings import Oracle, Reference
    lib = Reference() if Reference.available() else Oracle()
    F = batch.fft_size
    n_cpu = min(n_cpu, len(xs))
    t_f0, t_sp, t_ap, t_y = outs[1].cpu().numpy(), outs[2].cpu().numpy(), outs[3].cpu().numpy(), y.cpu().numpy()
    fo, yo = batch.frame_offsets, batch.out_offsets
    frames = 0
    df0 = 0.0
    se_sp = se_ap = 0.0
    cnt = 0
    dy = 0.0
    t0 = time.perf_counter()
    cpu_time = 0.0
    for u in range(n_cpu):
        x = xs[u]
        a = time.perf_counter()
        t, f0 = lib.dio(x, fs, fp)
        f0 = lib.stonemask(x, fs, t, f0)
        sp = lib.cheaptrick(x, fs, t, f0, -0.15, F)
        ap = lib.d4c(x, fs, t, f0, F, 0.0)
        yy = lib.synthesis(f0, sp, ap, F, fp, fs)
        cpu_time += time.perf_counter() - a
        frames += len(f0)
        g = slice(fo[u], fo[u + 1])
        df0 = max(df0, float(np.abs(t_f0[g] - f0).max()))
        se_sp += float(((t_sp[g] - sp) ** 2).sum())
        se_ap += float(((t_ap[g] - ap) ** 2).sum())
        cnt += sp.size
        dy = max(dy, float(np.abs(t_y[yo[u]:yo[u + 1]] - yy).max()))
        if time.perf_counter() - t0 > 40.0:
            n_cpu = u + 1
            break
    cpu = {"value": round(frames / cpu_time, 1), "unit": "frames/s", "cores": 1, "kind": lib.kind,
           "sample": "first %d utterances of the same batch (%d frames), Dio+StoneMask+CheapTrick+D4C+Synthesis, "
                     "single thread, %.1f s" % (n_cpu, frames, cpu_time)}
    parity = {"vs": lib.kind, "utterances": n_cpu, "max_abs_dF0_hz": df0, "sp_rmse": (se_sp / cnt) ** 0.5,
              "ap_rmse": (se_ap / cnt) ** 0.5, "max_abs_dy": dy}
    return cpu, parity


if __name__ == "__main__":
    main()
