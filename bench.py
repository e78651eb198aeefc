#!/usr/bin/env python3
"""bench.py -- WORLD analysis+synthesis frames/sec on MI355X (BASELINE.json's metric).

One "step" = one pass of the hot path over one batch of synthetic 16 kHz utterances that is
already resident in HBM: Dio -> StoneMask -> CheapTrick -> D4C (analysis) then Synthesis from the
features just produced, every utterance of the batch, fp64 as the reference computes.  Workload at
N=1 is BASELINE.json configs[1]: 256 synthetic utterances of 2-8 s.  With --gpus N every rank owns
its own 256 utterances (weak scaling, no data-path collective); at N > 1 the line also carries
`with_gather`: the same step followed by the RCCL gather-v of the float32 feature files to rank 0.

Prints ONE JSON line on rank 0 (contract in the task description) with these extra objects:
  roofline       -- dominant kernel (d4c_kernel): algorithmic bytes per launch / HIP-event duration
                    measured on the launch stream, against the 8 TB/s HBM peak; the FP64 figures that
                    actually bound the path are reported beside it.
  cpu_baseline   -- the reference WORLD (oracle/_ref, kind "reference") or this repo's C restatement
                    (kind "port") timed single-threaded on a bounded sample of the same workload.
  side_workloads -- (N = 1, default run) compact lines of the other BASELINE.json configurations, measured in the
                    same process after the headline: configs[2] Harvest, configs[4] Synthesis only, configs[3]
                    the corpus sweep; each with its own roofline / cpu_baseline / parity.  `--workload X` runs
                    one of them alone and prints its full line.
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
      d4c_kernel: its hop new samples + t + f0 + ap0 in, one ap row out (4768)
      pulse     : synth_pulse_kernel: f0 + one sp row + one ap row in per frame (its responses are intermediates,
                  y is the overlap-add kernel's)                       (8216)"""
    hop = fs * fp_ms / 1000.0
    bins = fft_size // 2 + 1
    analysis = hop * 8 + 16 + 2 * bins * 8
    synthesis = 8 + 2 * bins * 8 + hop * 8
    d4c = hop * 8 + 24 + bins * 8
    return {"analysis": analysis, "synthesis": synthesis, "round_trip": analysis + synthesis, "d4c": d4c,
            "pulse": 8 + 2 * bins * 8}


def d4c_flops_per_voiced_frame(fs):
    """Real FFTs of the D4C body per voiced frame: 2 x 2 (centroids) + 1 (power) + one per band, each
    2.5 N log2 N (scans, windows and the rest are not counted)."""
    fd = 2 ** (1 + int(np.log2(4.0 * fs / 47.0 + 1)))                       # d4c.cpp:344-346
    bands = int(min(15000.0, fs / 2.0 - 3000.0) / 3000.0)                  # d4c.cpp:351-353
    return (5 + bands) * 2.5 * fd * np.log2(fd)


# HBM-side bytes of a workload's dominant kernel come from PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
# runs, tools/profile_round.sh) stored in profiles/pmc_traffic.json together with a hash of the kernel's sources: when
# the sources have changed since the counters were collected the figure is stale and `traffic` is reported as null.
# The counter readings are corrected by the factors measured with tools/micro/hbmcal.hip on this chip
# (profiles/hbm_counter_calibration.json; MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes read).
COMMON_SOURCES = ("fft.hpp", "common.hpp", "wavesync.hpp", "partition.hpp", "fastmath.hpp")
KERNEL_SOURCES = {
    "analysis_synthesis": ("d4c.hip", "d4c_big.hpp", "peel.hpp", "frame.hpp", "spectrum.hpp", "window.hpp") + COMMON_SOURCES,
    "harvest": ("harvest.hip", "fftconv.hpp", "zcfilter.hpp", "decimate.hpp") + COMMON_SOURCES,
    "synthesis": ("synthesis.hip", "frame.hpp", "spectrum.hpp", "window.hpp") + COMMON_SOURCES,
}
KERNEL_SOURCES["sweep"] = KERNEL_SOURCES["analysis_synthesis"]


def kernel_source_hash(workload="analysis_synthesis"):
    import hashlib
    h = hashlib.sha256()
    for n in KERNEL_SOURCES[workload]:
        with open(os.path.join(ROOT, "hts-train-world_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def counter_calibration():
    """(fetch factor, write factor, note) from profiles/hbm_counter_calibration.json; (1, 1, why) without it."""
    path = os.path.join(ROOT, "profiles", "hbm_counter_calibration.json")
    try:
        with open(path) as f:
            c = json.load(f)
        return (float(c["fetch_factor_8B_per_lane"]), float(c["write_factor_8B_per_lane"]),
                "counters x (%.2f read, %.2f write): profiles/hbm_counter_calibration.json (tools/micro/hbmcal.hip, 8 B per lane)"
                % (c["fetch_factor_8B_per_lane"], c["write_factor_8B_per_lane"]))
    except (OSError, ValueError, KeyError):
        return 1.0, 1.0, "uncalibrated counters (profiles/hbm_counter_calibration.json missing)"


def measured_traffic(fs, workload="analysis_synthesis"):
    """(bytes per frame, note) from profiles/pmc_traffic.json, or (None, reason)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            recs = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/pmc_traffic.json missing"
    key = "analysis_synthesis" if workload == "sweep" else workload
    sha = kernel_source_hash(key)
    mine = [r for r in recs if r.get("fs") == fs and r.get("workload", "analysis_synthesis") == key]
    if not mine:
        return None, "no PMC passes recorded for %s at fs %d in profiles/pmc_traffic.json" % (key, fs)
    ff, wf, cal = counter_calibration()
    for r in mine:
        if r.get("source_sha") == sha:
            per = (r["fetch_kb"] * ff + r["write_kb"] * wf) * 1024.0 / r["frames"]
            return per, "PMC FETCH_SIZE + WRITE_SIZE of %s, separate passes (%s), %d frames; %s" % (
                r["kernel"], r.get("files", "profiles/"), r["frames"], cal)
    return None, "kernel sources changed since the PMC passes in profiles/pmc_traffic.json (now %s)" % sha


# timing scope of the library (WorldMi355TimingQuery) -> kernel names of the PMC passes it brackets
SCOPE_KERNELS = {
    "dio_lowcut_kernel": ("dio_lowcut_fft_kernel",),
    "dio_band_kernel": ("dio_band_fft_kernel", "dio_band_scan_kernel", "dio_band_compact_kernel"),
    "d4c_lovetrain_kernel": ("d4c_lovetrain_all_pass_kernel", "d4c_lovetrain_kernel", "d4cb_lovetrain_kernel"),
    "d4c_kernel": ("d4c_kernel", "d4cb_centroid_kernel", "d4cb_spectrum_kernel", "d4cb_band_kernel", "d4cb_output_kernel"),
    "synth_search_kernel": ("synth_pulse_search_kernel",),
    "hv_decimate": ("decim_fwd_kernel", "decim_bwd_kernel"),
    "hv_band_kernel": ("hv_band_fft_kernel", "hv_band_scan_kernel", "hv_band_compact_kernel"),
}


def per_kernel_rooflines(fs, workload, kernel_ms, steps, units):
    """`roofline.kernels`: every timed scope of the workload with its HIP-event time per step, the bytes it moved at
    the L2's memory side (the PMC passes of profiles/pmc_traffic.json, calibrated, scaled to this step's units) and
    what that is against the HBM peak.  None when the passes are not those of the current sources."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            recs = json.load(f)
    except (OSError, ValueError):
        return None
    key = "analysis_synthesis" if workload == "sweep" else workload
    sha = kernel_source_hash(key)
    rec = next((r for r in recs if r.get("fs") == fs and r.get("workload", "analysis_synthesis") == key
                and r.get("source_sha") == sha and r.get("per_kernel")), None)
    if rec is None:
        return None
    ff, wf, _ = counter_calibration()
    out = {}
    for scope, (ms, n) in kernel_ms.items():
        if n == 0:
            continue
        names = SCOPE_KERNELS.get(scope, (scope,))
        kb = sum(rec["per_kernel"][k]["fetch_kb"] * ff + rec["per_kernel"][k]["write_kb"] * wf
                 for k in names if k in rec["per_kernel"])
        per_step_ms = ms / max(1, steps)
        moved = kb * 1024.0 * units / rec["frames"]
        gbs = moved / (per_step_ms * 1e-3) / 1e9 if per_step_ms > 0 else None
        out[scope] = {"ms": round(per_step_ms, 4), "moved_mb": round(moved / 1e6, 1),
                      "gbs": round(gbs, 1) if gbs is not None else None,
                      "frac_hbm": round(gbs / HBM_PEAK_GBS, 4) if gbs is not None else None}
    return out


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
    ap.add_argument("--gather", action="store_true", help="include the RCCL feature gather in the timed step itself")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-inclusive", action="store_true",
                    help="skip the two PCIe-inclusive legs: every launch of the run is then one the line's launch_ms "
                         "averages (what tools/profile_round.sh profiles)")
    ap.add_argument("--detail-file", default=None,
                    help="where the FULL line goes (per-kernel tables, notes); default gpurun_out/bench_detail.json "
                         "when that directory exists.  stdout carries the compact line (driver_line()).")
    ap.add_argument("--full-line", action="store_true", help="print the full line on stdout instead of the compact one")
    ap.add_argument("--no-side", action="store_true",
                    help="analysis_synthesis at one GPU: skip the side_workloads (configs[2], [4], [3]) after the headline")
    ap.add_argument("--separate-calls", action="store_true",
                    help="WorldMi355Analyze then WorldMi355Synthesis instead of WorldMi355AnalyzeSynthesize")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl (= RCCL; the driver's launch) or gloo: rehearsal of the N > 1 control path with several "
                         "ranks sharing one GPU, where RCCL refuses duplicate devices")
    ap.add_argument("--cpu-utts", type=int, default=48, help="utterances of the CPU baseline sample")
    ap.add_argument("--workers", type=int, default=0, help="processes for synthetic data generation (0 = auto)")
    ap.add_argument("--workload", choices=["analysis_synthesis", "sweep", "harvest", "synthesis", "codec"],
                    default="analysis_synthesis",
                    help="analysis_synthesis = configs[1] (the headline metric, followed at one GPU by the side workloads); "
                         "sweep = configs[3]: a fixed corpus of ~1000 utterances sharded over the ranks (LPT), analysed, "
                         "gathered to rank 0 as float32 and written to files by rank 0 (strong scaling); harvest = configs[2] "
                         "(48 kHz, 1 ms, 64 utterances); synthesis = configs[4] (Synthesis only from precomputed features); "
                         "codec = the recipe's coded lf0/mgc/bap from resident features plus the decoders (SURVEY.md 8(f))")
    ap.add_argument("--raw", action="store_true",
                    help="sweep: write raw float32 f0/sp/ap (4.1 KB per frame) instead of what the recipe's own call writes "
                         "(data/Makefile.in:214: coded lf0 / mgc[50] / bap[25], 304 B per frame)")
    ap.add_argument("--coded", action="store_true", help="sweep: the default since round 3 (kept for older command lines)")
    ap.add_argument("--rounds", type=int, default=0,
                    help="sweep: batches per rank; one round is gathered, copied and written while the next is analysed.  "
                         "0 = auto: 2 to 4, keeping a rank's batch at 130 k frames or more -- smaller batches cost more than "
                         "the shorter exposed round of rank 0's copy + write gains (tools/sweep_knobs.sh on one GPU: 42.8 / "
                         "43.6 / 51.6 ms per pass at 4 / 8 / 16 rounds)")
    ap.add_argument("--writers", choices=("rank0", "all"), default="rank0",
                    help="sweep: rank0 = configs[3] as stated (gather-v, rank 0 writes everything); all = no gather, every "
                         "rank writes its own shard's files (shows what the rank-0 funnel costs)")
    ap.add_argument("--io-threads", type=int, default=0,
                    help="sweep: file-writing threads on the writing rank (0 = two per usable host core -- the cgroup's "
                         "share --, 4..64; with --writers all the cores are shared out over the ranks)")
    ap.add_argument("--plan-only", action="store_true",
                    help="print this rank's share of the workload as JSON and exit without touching the GPU")
    ap.add_argument("--out-dir", default=None, help="sweep: where rank 0 writes the feature files (default: a fresh temp dir, removed afterwards)")
    args = ap.parse_args()
    if args.utts <= 0:
        args.utts = {"sweep": 1000, "harvest": 64, "synthesis": 1024}.get(args.workload, 256)
    args.coded = not args.raw
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


class Env:
    """What every workload needs: torch, the package, rank layout, barrier and reductions over ranks."""

    def __init__(self, args, torch, dist, pkg, rank, world, shared_gpu):
        self.args, self.torch, self.dist, self.pkg = args, torch, dist, pkg
        self.W, self.sd, self.sh = pkg.world, pkg.synth_data, pkg.sharding
        self.rank, self.world, self.shared_gpu = rank, world, shared_gpu

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, elapsed, frames):
        """(max elapsed over ranks, total frames over ranks)."""
        if self.world == 1:
            return elapsed, float(frames)
        torch, dist = self.torch, self.dist
        tt = torch.tensor([elapsed, float(frames)], dtype=torch.float64,
                          device="cuda" if self.args.backend == "nccl" else "cpu")
        tmax, tsum = tt.clone(), tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        return float(tmax[0]), float(tsum[1])

    def comm_record(self):
        """What the communicator itself reports at N > 1 (every rank calls this; rank 0 keeps the result): backend,
        the world size as torch.distributed sees it, the RCCL version, and one entry per rank -- device index, name,
        PCI bus id -- gathered THROUGH the communicator, so that a scaling record proves N ranks on N devices took part."""
        if self.world == 1:
            return None
        torch, dist = self.torch, self.dist
        dev = torch.cuda.current_device()
        prop = torch.cuda.get_device_properties(dev)
        mine = {"rank": self.rank, "device": dev, "name": prop.name,
                "pci_bus_id": getattr(prop, "pci_bus_id", None), "host": os.uname().nodename}
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine)
        # a collective that every rank must take part in: the sum of (rank + 1) over the communicator
        chk = torch.tensor([float(self.rank + 1)], dtype=torch.float64,
                           device="cuda" if self.args.backend == "nccl" else "cpu")
        dist.all_reduce(chk)
        ver = None
        if self.args.backend == "nccl":
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:                       # noqa: BLE001 -- a record, not a dependency
                ver = "unavailable: %s" % e
        return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "version": ver,
                "library": "RCCL (torch.distributed backend nccl on ROCm)" if self.args.backend == "nccl" else "gloo (rehearsal)",
                "all_reduce_of_rank_plus_1": float(chk[0]), "expected": self.world * (self.world + 1) / 2.0,
                "devices": everyone}

    def rehearsal_note(self):
        if self.shared_gpu or (self.world > 1 and self.args.backend == "gloo"):
            return "rehearsal: %d ranks over gloo on %d GPU(s)" % (self.world, self.torch.cuda.device_count())
        return None


def timed_steps(env, step, steps, warmup, prewarm, ramp_step=None):
    """The contract's timing: untimed warm-up, then exactly `steps` steps between barrier + synchronize on both sides.
    Returns (elapsed seconds on this rank, clock_ramp object or None)."""
    ramp = clock_ramp(ramp_step or step, env.barrier, prewarm, steps) if prewarm > 0 else None
    for _ in range(warmup):
        step()
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    env.barrier()
    return time.perf_counter() - t0, ramp


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
    sd, sh = pkg.synth_data, pkg.sharding

    # ---- synthetic workload (not timed).  Generated BEFORE the GPU is touched: the generator forks a
    # worker pool, and a process that has initialised HIP / RCCL should not be forked ----
    fs, fp, utts = workload_spec(args)
    ncpu = os.cpu_count() or 1
    workers = args.workers if args.workers > 0 else max(1, min(16, ncpu // max(1, world)))
    if args.workers <= 0 and any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES")):
        workers = 1      # under rocprofv3 the preloaded tool has already initialised the GPU: do not fork
    side = (args.workload == "analysis_synthesis" and world == 1 and not args.no_side and args.fs == 16000
            and workers > 1)
    plan = None
    side_data = {}
    if args.workload == "sweep":
        # a FIXED corpus whatever the number of ranks (strong scaling): every rank derives the same partition from
        # the lengths alone and generates only its own utterances
        counts = [sd.utterance_samples(i, fs, tuple(args.dur)) for i in range(utts)]
        frames_all = [sh.frame_count(n, fs, fp) for n in counts]
        mine = sh.lpt_shards(frames_all, world, sh.rank0_handicap(world) if args.writers == "rank0" else None)[rank]
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
        if side:
            # configs[2]: 64 utterances at 48 kHz; configs[3]: the corpus of 1000 (its first `utts` are the headline's)
            side_data["harvest"] = sd.make_batch(64, 48000, (2.0, 8.0), workers=workers)
            side_data["corpus"] = xs + sd.make_batch(1000 - utts, fs, (2.0, 8.0), first=utts, workers=workers) \
                if utts < 1000 else xs[:1000]
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
    env = Env(args, torch, dist, pkg, rank, world, shared_gpu)
    ctx = pkg.world.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)

    if args.workload == "sweep":
        by_id = dict(zip(plan[1], xs))
        line = sweep_bench(env, ctx, plan[0], by_id, fs, fp, args.steps, args.warmup, cpu=not args.no_cpu_baseline)
    elif args.workload == "harvest":
        line = harvest_bench(env, ctx, xs, fs, fp, args.steps, args.warmup, args.prewarm, cpu=not args.no_cpu_baseline)
    elif args.workload == "synthesis":
        line = synthesis_bench(env, ctx, xs, None, fs, fp, args.steps, args.warmup, args.prewarm,
                               cpu=not args.no_cpu_baseline)
    elif args.workload == "codec":
        line = codec_bench(env, ctx, xs, fs, fp, args.steps, args.warmup, args.prewarm, cpu=not args.no_cpu_baseline)
    else:
        line = headline_bench(env, ctx, xs, fs, fp, cpu_all, side_data if side else None)
    if rank == 0 and line is not None:
        detail = write_detail(args, line)
        print(json.dumps(line if args.full_line else driver_line(line, detail)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
# configs[1]: the headline
# ------------------------------------------------------------------------------------------------------------------
def headline_bench(env, ctx, xs, fs, fp, cpu_all, side_data):
    args, torch, W, sh = env.args, env.torch, env.W, env.sh
    rank, world = env.rank, env.world
    lens = [len(x) for x in xs]
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    batch = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=lens)
    frames = int(batch.total_frames)
    outs = (torch.empty(frames, dtype=torch.float64, device="cuda"), torch.empty(frames, dtype=torch.float64, device="cuda"),
            torch.empty(frames, batch.bins, dtype=torch.float64, device="cuda"),
            torch.empty(frames, batch.bins, dtype=torch.float64, device="cuda"))
    y = torch.empty(int(batch.total_out), dtype=torch.float64, device="cuda")
    frame_counts = np.diff(batch.frame_offsets).tolist()
    all_counts = None
    if world > 1:                                   # every rank's frame counts, once: the gather's layout
        all_counts = [None] * world
        env.dist.all_gather_object(all_counts, [int(c) for c in frame_counts])
    bm = byte_model(fs, fp, batch.fft_size)
    # configs[3] (the corpus sweep) is measured FIRST when it rides along: as the last side workload of the process it
    # took 46-47.7 ms per pass where `--workload sweep` alone takes 40-41 (its last round's copy + file writes, the tail
    # no compute hides, 8 ms instead of 3.3: rounds 3-4, "unresolved"); as the first it is the run-alone pass.
    # WM_SWEEP_LAST=1 restores the old order (A/B).
    sides_early = {}
    pipes = {}
    if side_data is not None and rank == 0 and not os.environ.get("WM_SWEEP_LAST"):
        if not args.no_host_inclusive:
            pipes = build_host_pipelines(env, ctx, xs, fs, fp)
        corpus = side_data["corpus"]
        sides_early["sweep"] = compact(sweep_bench(env, ctx, [len(v) for v in corpus], dict(enumerate(corpus)), fs, fp, 3, 2,
                                                   cpu=not args.no_cpu_baseline))
        torch.cuda.empty_cache()

    def compute():
        if args.separate_calls:
            t, f0, sp, ap = batch.analyze(x, out=outs)
            batch.synthesize(f0, sp, ap, out=y)
        else:
            # the same launches as one call: Synthesis' f0-only first part runs beside CheapTrick / D4C
            t, f0, sp, ap, _ = batch.analyze_synthesize(x, out=outs, y=y)
        return f0, sp, ap

    def gather(f0, sp, ap, raw=False):
        # What travels to rank 0 is what its files hold.  Default: the recipe's own call (`analysis ... 5 F 50 25`,
        # data/Makefile.in:214): float32 lf0 / mgc[50] / bap[25], 304 B per frame, coded on every rank before the send
        # (WorldMi355RecipeFeatures).  raw: the CLI without compression arguments: float32 f0 / sp / ap
        # (test/analysis.cpp:360-390), 4.1 KB per frame -- seven peers' worth of that into one rank per step is what
        # no link budget scales.
        feats = list(batch.recipe_features(f0, sp, ap, 50, 25)) if not raw else [f0.float(), sp.float(), ap.float()]
        if args.backend == "gloo":
            feats = [v.cpu() for v in feats]
        sh.gather_features(feats, frame_counts, dst=0, all_counts=all_counts)

    def step():
        f0, sp, ap = compute()
        if args.gather and world > 1:
            gather(f0, sp, ap, raw=args.raw)

    # the ramp runs the rank-local part only: its length is set by each rank's clock, so it must not hold a collective
    ctx.timing_enable(False)
    ramp = clock_ramp(compute, env.barrier, args.prewarm, args.steps) if args.prewarm > 0 else None
    for _ in range(args.warmup):
        step()
    env.barrier()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    env.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = {k: ctx.timing_query(k) for k in ANALYSIS_KERNELS + SYNTHESIS_KERNELS}
    ctx.timing_enable(False)
    elapsed_max, total_frames = env.reduce(elapsed, frames)
    with_gather = with_gather_raw = None
    if world > 1 and not args.gather:
        # the same step followed by the one exchange the path has -- the gather-v of every rank's feature slabs to
        # rank 0 (RCCL grouped send/recv over xGMI) -- under the same timing contract: coded (what the recipe writes)
        # and raw
        def gather_leg(raw):
            def step_g():
                gather(*compute(), raw=raw)
            step_g()
            env.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_g()
            env.barrier()
            eg, _ = env.reduce(time.perf_counter() - t0, frames)
            per_rank = 4 * frames * ((1 + 2 * batch.bins) if raw else (1 + 50 + 25))
            # the exchange alone (features already computed): what the links into rank 0 carry
            f0_, sp_, ap_ = compute()
            gather(f0_, sp_, ap_, raw=raw)
            env.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                gather(f0_, sp_, ap_, raw=raw)
            env.barrier()
            ego, _ = env.reduce(time.perf_counter() - t0, frames)
            return {"value": round(total_frames * args.steps / eg, 1), "unit": "frames/s",
                    "ms_per_step": round(eg / args.steps * 1e3, 3),
                    "gathered_bytes_per_step": per_rank * (world - 1),
                    "gather_alone_ms": round(ego / args.steps * 1e3, 3),
                    "gather_alone_gbs_into_rank0": round(per_rank * (world - 1) * args.steps / ego / 1e9, 2),
                    "note": "analysis + synthesis, then %s of every rank gathered to rank 0 (%s); gather_alone includes "
                            "%s" % ("float32 f0/sp/ap" if raw else "the recipe's coded float32 lf0/mgc[50]/bap[25]",
                                    "RCCL send/recv" if args.backend == "nccl" else "gloo rehearsal",
                                    "the float64 -> float32 conversion of the slabs" if raw else "the coding kernels")}
        with_gather = gather_leg(False)
        with_gather_raw = gather_leg(True)
    comm = env.comm_record()
    hi = hic = None
    if not args.no_host_inclusive:
        hi = host_inclusive(env, ctx, xs, fs, fp, pipe=pipes.pop(None, None))
        hic = host_inclusive(env, ctx, xs, fs, fp, coded=(50, 25), pipe=pipes.pop((50, 25), None))

    line = None
    if rank == 0:
        value = total_frames * args.steps / elapsed_max
        voiced = int((outs[1] > 0).sum().item())
        roof = d4c_roofline(kernel_ms, frames, voiced, fs, bm, args.steps)
        roof["pipeline"] = {"achieved_gbs": round(value / world * bm["round_trip"] / 1e9, 3),
                            "frac_hbm": round(value / world * bm["round_trip"] / 1e9 / HBM_PEAK_GBS, 6),
                            "achieved_fp64_tflops": round(value / world * FLOPS_PER_FRAME / 1e12, 4) if fs == 16000 else None}
        cpu = None
        cpu_o2 = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, parity = cpu_baseline_and_parity(xs, fs, fp, batch, outs, y, args.cpu_utts)
            cpu_o2 = cpu_baseline_o2(xs, fs, fp, batch.fft_size, args.cpu_utts)
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
        if env.rehearsal_note():
            line["config"]["note"] = env.rehearsal_note()
        if with_gather:
            line["with_gather"] = with_gather
        if with_gather_raw:
            line["with_gather_raw"] = with_gather_raw
        if comm:
            line["rccl"] = comm
        if hi:
            line["host_inclusive"] = hi
        if hic:
            line["host_inclusive_coded"] = hic
        if cpu_o2:
            line["cpu_baseline_O2"] = cpu_o2
        if cpu_all:
            line["cpu_baseline_all_cores"] = cpu_all
        if parity:
            line["parity"] = parity
    # ---- the other BASELINE.json configurations, compact, in the same run (one GPU) ----
    if side_data is not None and rank == 0:
        feats = (outs[1].clone(), outs[2].clone(), outs[3].clone(), np.diff(batch.frame_offsets).tolist(),
                 np.diff(batch.out_offsets).tolist())
        del outs, y, x
        batch.close()
        torch.cuda.empty_cache()
        cpu = not args.no_cpu_baseline
        sides = dict(sides_early)
        corpus = side_data["corpus"]
        counts = [len(v) for v in corpus]
        for w in ("hs" if "sweep" in sides else "hsw"):
            if w == "h":
                sides["harvest"] = compact(harvest_bench(env, ctx, side_data["harvest"], 48000, 1.0, 5, 2, 2.0, cpu=cpu))
            elif w == "s":
                sides["synthesis"] = compact(synthesis_bench(env, ctx, xs, feats, fs, fp, 5, 2, 2.0, cpu=cpu))
            elif w == "w":
                # two warm-up passes, as `--workload sweep` takes: the second pass over a batch is the first in the
                # streamed order (D4C's preparation beside CheapTrick)
                sides["sweep"] = compact(sweep_bench(env, ctx, counts, dict(enumerate(corpus)), fs, fp, 3, 2, cpu=cpu))
            torch.cuda.empty_cache()
        del feats
        line["side_workloads"] = {k: sides[k] for k in ("harvest", "synthesis", "sweep") if k in sides}
    else:
        batch.close()
    return line


def compact(line):
    """A side workload's line reduced to what the headline line carries of it."""
    if line is None:
        return None
    roof = line.get("roofline") or {}
    keep_roof = {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_note",
                                          "launch_ms", "units_per_launch", "bytes_per_unit", "fp64", "kernel_ms_per_step", "kernels",
                                          "kernels_note")
                 if k in roof}
    out = {"metric": line["metric"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
           "steps": line["steps"], "scaling": line["scaling"], "dtype": line["dtype"],
           "config": {"workload": line["config"]["workload"], "frames": line["config"].get("frames_per_gpu",
                                                                                           line["config"].get("frames"))},
           "roofline": keep_roof, "cpu_baseline": line.get("cpu_baseline"), "parity": line.get("parity")}
    for k in ("phases_ms_per_step", "host_side", "predicted", "value_compute_only"):
        if k in line:
            out[k] = line[k]
    return out


# The line a driver records is the LAST 8 KB of stdout: round 4's full line (per-kernel tables and a paragraph of
# provenance per workload: 14 KB) lost its first half there.  stdout now carries driver_line(): every number of
# every workload, no table and no prose twice; the full line goes to a file beside it (write_detail).
DRIVER_LINE_MAX = 6500
_DROP_EVERYWHERE = ("kernels", "kernels_note", "traffic_note", "layout", "model", "flags", "overlapped")
_NOTE_KEYS = ("note",)


def _sig(v, digits=8):
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, v))
    return v


def _prune(o, depth=0, keep_notes=False):
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if k in _DROP_EVERYWHERE or (k in _NOTE_KEYS and not keep_notes):
                continue
            if k == "sample" and isinstance(v, str) and len(v) > 96:
                v = v[:93] + "..."
            out[k] = v if k in ("value", "ms_per_step") and not isinstance(v, (dict, list)) else _prune(v, depth + 1, keep_notes)
        return out
    if isinstance(o, list):
        return [_prune(v, depth + 1, keep_notes) for v in o]
    return _sig(o)


def driver_line(full, detail=None):
    """The compact stdout line: the contract's keys, `roofline` and `cpu_baseline` of the headline in full (numbers),
    every side workload's value / ms_per_step / roofline numbers / cpu_baseline / parity, the host-inclusive and
    gather figures -- and, last, `summary`: the handful of numbers a reader wants first."""
    line = _prune(full)
    cfg = dict(line.get("config") or {})
    if isinstance(cfg.get("workload"), str) and len(cfg["workload"]) > 150:
        cfg["workload"] = cfg["workload"][:147] + "..."
    line["config"] = cfg
    roof = (full.get("roofline") or {})
    notes = {}
    if roof.get("traffic_note"):
        notes["traffic"] = roof["traffic_note"] if len(roof["traffic_note"]) < 330 else roof["traffic_note"][:327] + "..."
    if roof.get("note"):
        notes["roofline"] = roof["note"]
    if detail:
        notes["detail_file"] = detail + " (full line: per-kernel MB moved / GB/s tables, provenance notes)"
    sides = {}
    for name, sl in (line.pop("side_workloads", None) or {}).items():
        if sl is None:
            continue
        for k in ("metric", "unit", "steps", "dtype"):
            sl.pop(k, None)
        sl["frames"] = (sl.pop("config", None) or {}).get("frames")
        r = sl.get("roofline") or {}
        if "fp64" in r:
            r["fp64_frac"] = (r.pop("fp64") or {}).get("frac")
        for k in ("peak", "unit", "launches_per_step"):
            r.pop(k, None)
        cb = sl.get("cpu_baseline")
        if cb:
            sl["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
        sides[name] = sl
    if sides:
        line["side_workloads"] = sides
    if notes:
        line["notes"] = notes
    summary = {"value": line.get("value"), "ms_per_step": line.get("ms_per_step"),
               "roofline_frac": (line.get("roofline") or {}).get("frac"),
               "d4c_launch_ms": (line.get("roofline") or {}).get("launch_ms")}
    for k in ("host_inclusive", "host_inclusive_coded", "with_gather", "with_gather_raw"):
        if isinstance(line.get(k), dict):
            summary[k] = line[k].get("value")
    for name, sl in sides.items():
        summary[name] = sl.get("value")
        summary[name + "_ms"] = sl.get("ms_per_step")
    if isinstance(line.get("cpu_baseline"), dict) and line["cpu_baseline"].get("value"):
        summary["x_cpu_1thread"] = _sig(line["value"] / line["cpu_baseline"]["value"], 4)
    if isinstance(full.get("parity"), dict):
        summary["parity"] = {k: _sig(v, 3) for k, v in full["parity"].items() if isinstance(v, float)}
    line["summary"] = summary
    text = json.dumps(line)
    if len(text) > DRIVER_LINE_MAX:
        # still too long (many ranks' rccl records, ...): drop the per-scope times of the side workloads, then the notes
        for sl in sides.values():
            (sl.get("roofline") or {}).pop("kernel_ms_per_step", None)
        if len(json.dumps(line)) > DRIVER_LINE_MAX:
            line.pop("notes", None)
        if len(json.dumps(line)) > DRIVER_LINE_MAX and isinstance(line.get("rccl"), dict):
            line["rccl"] = {k: v for k, v in line["rccl"].items() if k != "ranks"}
    return line


def write_detail(args, full):
    """The full line as a file; returns the path written (relative), or None."""
    path = args.detail_file
    if path is None:
        d = os.path.join(os.getcwd(), "gpurun_out")
        if not os.path.isdir(d):
            return None
        name = "bench_detail.json" if args.workload == "analysis_synthesis" and args.fs == 16000 else \
            "bench_detail_%s_%d.json" % (args.workload, args.fs)
        path = os.path.join("gpurun_out", name)
    try:
        with open(path, "w") as f:
            f.write(json.dumps(full) + "\n")
        return path
    except OSError:
        return None


def build_host_pipelines(env, ctx, xs, fs, fp):
    """The two host-inclusive legs' pipelines (pinned host buffers and device slots), built BEFORE anything else fills
    host memory and kept until their legs run.  Measured (profiles/r05_d_* .. r05_g_*): pinned memory obtained after the
    sweep has pushed gigabytes of feature files through the page cache downloads at 36 GB/s instead of 52 (host_inclusive
    7.7-8.4 M instead of 12.1-12.5 M frames/s), also when the same sizes had been allocated and freed before."""
    pkg = env.pkg
    return {coded: pkg.pipeline.HostPipeline(ctx, pkg.world.default_params(fs, fp), [len(x) for x in xs], synthesis=True,
                                             coded=coded)
            for coded in (None, (50, 25))}


def host_inclusive(env, ctx, xs, fs, fp, coded=None, pipe=None):
    """The same step fed from and drained to pinned HOST memory in the on-disk types (int16 samples up; float32
    f0 / sp / ap and int16 resynthesised samples down), double-buffered on copy streams beside the kernels
    (hts-train-world_amd/pipeline.py).  Reported beside `value`, never as it (the metric
    "including H2D of waveforms and D2H of features")."""
    args, torch, pkg = env.args, env.torch, env.pkg
    pl = pkg.pipeline
    if pipe is None:
        pipe = pl.HostPipeline(ctx, pkg.world.default_params(fs, fp), [len(x) for x in xs], synthesis=True, coded=coded)
    x16 = pl.to_int16(np.concatenate(xs))
    for xb in pipe.x_pinned:                        # the waveforms wait in pinned memory, as decoded wav payloads would
        xb.numpy()[:] = x16
    steps = max(8, args.steps)                      # a pipeline: its fill and drain are amortised over the steps
    for _ in range(2):
        pipe.result(pipe.submit())
    env.barrier()
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
    dt_max, total = env.reduce(dt, frames)
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
HARVEST_KERNELS = ("hv_decimate", "hv_band_kernel", "hv_band_fft_kernel", "hv_raw_kernel", "hv_detect_kernel",
                   "hv_refine_kernel", "hv_remove_kernel", "hv_contour_kernel")


def kernel_roofline(workload, kernel, note, kernel_ms, steps, units, bytes_per_unit, fs, flops_per_launch=None):
    """The `roofline` object of a workload's dominant kernel: algorithmic bytes per launch over the HIP-event duration
    of its launches (recorded on the launch stream), against the HBM peak; FP64 figures beside it."""
    ms, n = kernel_ms[kernel]
    avg_s = ms / max(1, n) * 1e-3
    per_step = max(1, round(n / max(1, steps)))        # a step may launch the kernel several times (batches)
    units = units / per_step
    per_unit, tnote = measured_traffic(fs, workload)
    ok = avg_s > 0
    roof = {
        "bound": "hbm", "kernel": kernel,
        "achieved": round(units * bytes_per_unit / avg_s / 1e9, 3) if ok else None,
        "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(units * bytes_per_unit / avg_s / 1e9 / HBM_PEAK_GBS, 6) if ok else None,
        "traffic": round(units * per_unit) if per_unit is not None else None,
        "launches_per_step": per_step, "traffic_note": tnote,
        "launch_ms": round(avg_s * 1e3, 4), "units_per_launch": round(units), "bytes_per_unit": bytes_per_unit,
        "note": note,
        "kernel_ms_per_step": {k: round(v[0] / steps, 4) for k, v in kernel_ms.items()},
    }
    if flops_per_launch is not None:
        fl = flops_per_launch / per_step
        roof["fp64"] = {"achieved_tflops": round(fl / avg_s / 1e12, 3) if ok else None, "peak_tflops": FP64_PEAK_TFLOPS,
                        "frac": round(fl / avg_s / 1e12 / FP64_PEAK_TFLOPS, 5) if ok else None}
    kernels = per_kernel_rooflines(fs, workload, kernel_ms, steps, units * per_step)
    if kernels:
        roof["kernels"] = kernels
        roof["kernels_note"] = ("per timed scope: HIP-event ms per step, MB moved at the L2's memory side (PMC FETCH_SIZE x read "
                                "factor + WRITE_SIZE per kernel, %s), GB/s and fraction of the %g GB/s HBM peak; scopes on "
                                "different streams overlap in time" % (tnote.split(";")[0] if tnote else "", HBM_PEAK_GBS))
    return roof


def d4c_roofline(kernel_ms, frames, voiced, fs, bm, steps, workload="analysis_synthesis"):
    fd = 2 ** (1 + int(math.log2(4.0 * fs / 47.0 + 1.0)))          # fft_size_d4c (d4c.cpp:341-343)
    name = "d4c_kernel" if fd <= 2048 else "d4c_kernel scope = d4cb_centroid + d4cb_spectrum + d4cb_band + d4cb_output (fft %d)" % fd
    roof = kernel_roofline(workload, "d4c_kernel", "FP64-FFT/LDS bound, not HBM bound (SURVEY.md 8d); fp64 figures beside it",
                           kernel_ms, steps, frames, bm["d4c"], fs, flops_per_launch=voiced * d4c_flops_per_voiced_frame(fs))
    roof["kernel"] = name
    return roof


# ------------------------------------------------------------------------------------------------------------------
# configs[3]: the corpus sweep
# ------------------------------------------------------------------------------------------------------------------
def sh_frames(n, fs, fp):
    return int(1000.0 * n / fs / fp) + 1                      # GetSamplesForDIO, dio.cpp:638-640


def sweep_bench(env, ctx, counts, by_id, fs, fp, steps, warmup, cpu=True):
    """configs[3]: the data/ feature-extraction sweep over a fixed corpus (data/Makefile.in:125-242), sharded over the
    ranks; one step = every rank analyses its shard, the float32 feature slabs are gathered to rank 0, rank 0 writes
    every utterance's files (what the recipe's call writes: coded lf0 / mgc / bap; --raw for f0 / sp / ap).  Waveforms
    are resident in HBM when the timed region starts.  by_id: this rank's utterances by corpus index."""
    import shutil
    import tempfile
    args, torch, dist, pkg = env.args, env.torch, env.dist, env.pkg
    rank, world = env.rank, env.world
    W, sweep = pkg.world, pkg.sweep
    rounds = args.rounds
    if rounds <= 0:                                   # auto (see --rounds)
        per_rank = sum(sh_frames(n, fs, fp) for n in counts) / float(world)
        rounds = int(max(2, min(4, per_rank // 130000)))
    sw = sweep.ShardedSweep(ctx, fs, fp, counts, rank, world, spec_dim=50 if args.coded else 0, ap_dim=25,
                            backend=args.backend, rounds=rounds, writers=args.writers)
    mine = sw.shards[rank]
    sw.load(lambda i: by_id[i])
    out_dir = None
    sink = None
    if rank == 0:
        # a RAM-backed directory when there is one: the container's overlay root creates and fills files at a
        # fraction of the rate (measured: 515 ms per raw pass there against 139 ms on /dev/shm), and it is the sweep
        # that is under test, not that filesystem; --out-dir names any other place
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        out_dir = args.out_dir or tempfile.mkdtemp(prefix="wm_sweep_", dir=shm)
    if args.writers == "all" and world > 1:
        box = [out_dir]
        dist.broadcast_object_list(box, src=0)
        out_dir = box[0]
    if out_dir:
        names = ("lf0", "mgc", "bap") if args.coded else ("f0", "sp", "ap")
        sink = sweep.dir_sink(out_dir, names)
    ncpu = pkg.sharding.usable_cpus()                 # the cgroup's share, not the host's core count
    io_threads = args.io_threads or max(4, min(64, 2 * ncpu if args.writers == "rank0" else 2 * ncpu // max(1, world)))
    comm = env.comm_record()

    for _ in range(warmup):
        sw.run(sink, io_threads)
    env.barrier()
    ctx.timing_enable(True)
    phases = {"compute": 0.0, "gather": 0.0, "to_host": 0.0, "write": 0.0, "wall": 0.0}
    t0 = time.perf_counter()
    for _ in range(steps):
        ph = sw.run(sink, io_threads)
        if os.environ.get("WM_BENCH_PHASES"):
            print("  pass: %s" % {k: round(v * 1e3, 2) for k, v in ph.items() if isinstance(v, float)}, file=sys.stderr)
        for k in phases:
            phases[k] += ph.get(k, 0.0)
    env.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("WM_BENCH_PHASES"):
        print("sweep phases (s over %d steps): %s elapsed %.4f" % (steps, {k: round(v, 4) for k, v in phases.items()}, elapsed), file=sys.stderr)
    kernel_ms = {k: ctx.timing_query(k) for k in ANALYSIS_KERNELS}
    ctx.timing_enable(False)
    elapsed_max, _ = env.reduce(elapsed, sw.my_frames)
    # slowest rank's compute, the busiest rank's share of frames (LPT balance)
    comp_max, _ = env.reduce(phases["compute"], 0)
    most, _ = env.reduce(float(sw.my_frames), 0)
    line = None
    if rank == 0:
        total = sw.total_frames
        value = total * steps / elapsed_max
        F = pkg.capi.cheaptrick_fft_size(fs)
        bm = byte_model(fs, fp, F)
        frames0 = sum(int(b.total_frames) for b, _ in sw.loaded)
        voiced = 0
        for b, xx in sw.loaded:
            voiced += int((b.analyze(xx)[1] > 0).sum().item())
        roof = d4c_roofline(kernel_ms, frames0, voiced, fs, bm, steps, workload="sweep") if frames0 else None
        per_frame_out = 4 * ((1 + 50 + 25) if args.coded else (1 + 2 * (F // 2 + 1)))
        k = float(steps)
        ms = lambda v: round(v / k * 1e3, 3)
        # What eight ranks would take, from this run's phases: compute divides by the ranks; the gather, the copy to the
        # host and the file writes stay with rank 0 (its xGMI links carry 7/8 of the bytes the copy engine then moves
        # down).  A pass is a pipeline over the rounds: the longest stage bounds it, one round of the others is exposed.
        R = max(1, sw.rounds)
        h8 = pkg.sharding.rank0_handicap(8)[0] if args.writers == "rank0" else 1.0
        c8 = comp_max / k / (7.0 + 1.0 / h8)             # the seven peers' share (rank 0's is smaller by its handicap)
        host_side = (phases["to_host"] + phases["gather"]) / k
        wr = phases["write"] / k
        n_files = 3 * len(counts)

        def pass8(cores):
            # the file writes are page-cache fills by two native threads per usable core: they scale with the cores
            # of the node the eight ranks run on (rank 0's process sees all of them there), the PCIe copy does not
            w = wr * float(ncpu) / float(cores)
            longest = max(c8, host_side, w)
            return longest + (c8 + host_side + w - longest) / R
        pred8 = pass8(ncpu)

        def pass8_all(cores):
            # --writers all: no funnel -- every rank copies and writes its own eighth (the ranks share the node's cores)
            ca = comp_max / k / 8.0
            ha = phases["to_host"] / k / 8.0
            w = wr * float(ncpu) / float(cores)
            longest = max(ca, ha, w)
            return longest + (ca + ha + w - longest) / R
        by_cores_all = {str(c): round(pass8_all(c) * 1e3, 3) for c in (16, 64, 128)}
        by_cores = {str(c): round(pass8(c) * 1e3, 3) for c in (16, 64, 128)}
        speed_by_cores = {c: round(elapsed_max / steps / (v * 1e-3), 2) for c, v in by_cores.items()} if world == 1 else None
        line = {
            "metric": "WORLD analysis sweep frames/sec @16kHz, 5ms hop (corpus -> float32 feature files on rank 0)",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed_max / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[3]: fixed corpus of %d synthetic 16 kHz utterances (%g-%g s, %d frames), LPT-sharded "
                                   "over %d rank(s); Dio+StoneMask+CheapTrick+D4C in fp64, %s float32 gathered to rank 0 "
                                   "(grouped send/recv), written by rank 0 as one file per utterance and feature; rounds "
                                   "pipelined: analysis | gather + copy to host | file writes"
                                   % (len(counts), args.dur[0], args.dur[1], total, world,
                                      "coded lf0/mgc(50)/bap(25) (the recipe's call, data/Makefile.in:214)" if args.coded
                                      else "raw f0/sp/ap"),
                       "fs": fs, "frame_period_ms": fp, "utterances": len(counts), "frames": total,
                       "frames_on_busiest_rank": int(most), "output_bytes_per_frame": per_frame_out,
                       "rounds": sw.rounds, "writers": args.writers, "out_dir": os.path.dirname(out_dir) or out_dir,
                       "parallelism": "utterance-sharded x%d, %s" % (world, "gather-v to rank 0 over " + args.backend
                                                                      if args.writers == "rank0" else "every rank writes its shard")},
            "phases_ms_per_step": {"compute_slowest_rank": ms(comp_max), "gather_rank0": ms(phases["gather"]),
                                   "to_host_rank0": ms(phases["to_host"]), "file_write_rank0": ms(phases["write"]),
                                   "pass_wall_rank0": ms(phases["wall"]),
                                   "note": "busy times per stage (HIP events per stream; writes: time inside the native "
                                           "writer); they overlap inside pass_wall"},
            "host_side": {"usable_cpus": ncpu, "os_cpu_count": os.cpu_count(), "writer_threads": io_threads, "files_per_step": n_files,
                          "files_per_s": round(n_files / wr, 1) if wr > 0 else None,
                          "write_gbs": round(total * per_frame_out / wr / 1e9, 2) if wr > 0 else None,
                          "to_host_gbs": round(total * per_frame_out / (phases["to_host"] / k) / 1e9, 2)
                          if phases["to_host"] > 0 else None},
            "predicted": {"ranks": 8, "ms_per_step": round(pred8 * 1e3, 3),
                          "speedup_over_this_run": round(elapsed_max / steps / pred8, 2) if world == 1 and pred8 > 0 else None,
                          "host_cores_assumed": ncpu, "writer_threads_assumed": io_threads,
                          "ms_per_step_by_host_cores": by_cores, "speedup_by_host_cores": speed_by_cores,
                          "writers_all_ms_by_host_cores": by_cores_all,
                          "rank0_handicap": round(h8, 3),
                          "model": "max(compute/(7 + 1/handicap), gather + to_host, write) + (the other two stages) / "
                                   "rounds, from this run's phases; the writes are page-cache fills by two native threads "
                                   "per usable core and scale with the host cores rank 0 may use (this box: 16; an "
                                   "8-GPU node: 64-128), the PCIe copy does not"},
            "value_compute_only": round(total * k / comp_max, 1) if comp_max > 0 else None,
            "roofline": roof, "cpu_baseline": None,
        }
        if env.rehearsal_note():
            line["config"]["note"] = env.rehearsal_note()
        if comm:
            line["rccl"] = comm
        if world == 1 and cpu:
            line["cpu_baseline"], line["parity"] = sweep_cpu_baseline(by_id, mine, fs, fp, out_dir, args)
        if not args.out_dir:
            shutil.rmtree(out_dir, ignore_errors=True)
    sw.close()
    return line


def sweep_cpu_baseline(by_id, mine, fs, fp, out_dir, args):
    """The reference's `analysis` arithmetic (Dio+StoneMask+CheapTrick+D4C, float32 outputs; with the coder when the
    files are coded) single-threaded on a bounded sample of the corpus, and the files rank 0 wrote for those
    utterances checked against it."""
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
        if args.coded:                                             # analysis.cpp:292-366
            sp4 = sp * 1e4
            sp4[sp4 == 0.0] = 0.0001
            mgc = lib.code_spectral_envelope(sp4, fs, F, 50)
            mgc[:, 0] += 12.0
            bap = lib.code_spectral_envelope(ap * 1e4, fs, F, 25)
            bap[:, 0] -= 9.210340
            bap[(bap[:, 0] > 0) & (bap[:, 0] < 1e-4), 0] = 0.0
        tcpu += time.perf_counter() - a
        frames += len(f0)
        n += 1
        if args.coded:
            g_lf0 = np.fromfile(os.path.join(out_dir, "lf0", "utt%05d.lf0" % i), dtype=np.float32)
            g_mgc = np.fromfile(os.path.join(out_dir, "mgc", "utt%05d.mgc" % i), dtype=np.float32).reshape(-1, 50)
            g_bap = np.fromfile(os.path.join(out_dir, "bap", "utt%05d.bap" % i), dtype=np.float32).reshape(-1, 25)
            v = f0 > 0                                               # lf0 = 0 where unvoiced (analysis.cpp:216-224)
            assert np.array_equal(v, g_lf0 != 0), "voiced / unvoiced frames of utterance %d differ" % i
            if v.any():
                df0 = max(df0, float(np.abs(np.exp(g_lf0[v].astype(np.float64)) - f0[v]).max()))
            se_sp += float(((g_mgc.astype(np.float64) - mgc.astype(np.float32)) ** 2).sum())
            se_ap += float(((g_bap.astype(np.float64) - bap.astype(np.float32)) ** 2).sum())
            cnt += mgc.size
        else:
            g_f0 = np.fromfile(os.path.join(out_dir, "f0", "utt%05d.f0" % i), dtype=np.float32)
            g_sp = np.fromfile(os.path.join(out_dir, "sp", "utt%05d.sp" % i), dtype=np.float32).reshape(-1, F // 2 + 1)
            g_ap = np.fromfile(os.path.join(out_dir, "ap", "utt%05d.ap" % i), dtype=np.float32).reshape(-1, F // 2 + 1)
            df0 = max(df0, float(np.abs(g_f0 - f0.astype(np.float32)).max()))
            se_sp += float(((g_sp.astype(np.float64) - sp.astype(np.float32)) ** 2).sum())
            se_ap += float(((g_ap.astype(np.float64) - ap.astype(np.float32)) ** 2).sum())
            cnt += sp.size
        if tcpu > 12.0:
            break
    cpu = {"value": round(frames / tcpu, 1), "unit": "frames/s", "cores": 1, "kind": lib.kind,
           "sample": "%d utterances of the corpus (%d frames), Dio+StoneMask+CheapTrick+D4C%s, single thread, %.1f s "
                     "(file writing not included)" % (n, frames, " + the CLI's coding" if args.coded else "", tcpu)}
    parity = {"vs": lib.kind, "utterances": n,
              "files": "float32 %s as written by rank 0" % ("lf0/mgc/bap" if args.coded else "f0/sp/ap"),
              "max_abs_dF0_hz": df0, ("mgc_rmse" if args.coded else "sp_rmse"): (se_sp / max(1, cnt)) ** 0.5,
              ("bap_rmse" if args.coded else "ap_rmse"): (se_ap / max(1, cnt)) ** 0.5}
    return cpu, parity


def workload_spec(args):
    """(fs, frame period, utterances per GPU -- for the sweep: of the corpus) of the selected workload."""
    if args.workload == "harvest":
        return 48000, 1.0, args.utts
    if args.workload == "analysis_synthesis":
        return args.fs, 5.0, args.utts
    return 16000, 5.0, args.utts


# ------------------------------------------------------------------------------------------------------------------
# configs[2] (Harvest), configs[4] (Synthesis only), the feature codec
# ------------------------------------------------------------------------------------------------------------------
def side_line(env, metric, cfg, value, steps, warmup, elapsed, frames, roof, ramp):
    line = {"metric": "WORLD %s frames/sec" % metric, "value": round(value, 1), "unit": "frames/s", "n_gpus": env.world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": dict(cfg, frames_per_gpu=frames), "roofline": roof, "cpu_baseline": None}
    if ramp:
        line["clock_ramp"] = ramp
    if env.rehearsal_note():
        line["config"]["note"] = env.rehearsal_note()
    return line


def harvest_bench(env, ctx, xs, fs, fp, steps, warmup, prewarm, cpu=True):
    """configs[2]: Harvest at 48 kHz with a 1 ms hop over 64 utterances of 2-8 s."""
    torch, W = env.torch, env.W
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    batch = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(v) for v in xs])
    frames = int(batch.total_frames)
    step = lambda: batch.harvest(x)
    ctx.timing_enable(False)
    ramp = clock_ramp(step, env.barrier, prewarm, steps) if prewarm > 0 else None
    for _ in range(warmup):
        step()
    env.barrier()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    env.barrier()
    elapsed = time.perf_counter() - t0
    kms = {k: ctx.timing_query(k) for k in HARVEST_KERNELS}
    ctx.timing_enable(False)
    elapsed, total = env.reduce(elapsed, frames)
    line = None
    if env.rank == 0:
        value = total * steps / elapsed
        # The dominant kernel is the filter bank (hv_band_fft_kernel): per basic frame (1 ms = 8 samples of the 8 kHz
        # decimated signal) it must read those 8 samples (64 B; the event lists it writes are intermediates), and per
        # block of `step` samples it runs one forward and one inverse real transform of 2048 points per channel.
        r = max(1, min(12, int(fs / 8000.0 + 0.5)))
        nch = 1 + int(math.log2(800.0 * 1.1 / (71.0 * 0.9)) * 40)                 # harvest.cpp:1151-1153
        dec_per_frame = fs / r * fp / 1000.0
        flops = frames * dec_per_frame / 1790.0 * (nch + 1) * 2.5 * 2048 * 11      # blocks x (channels + 1) real FFT-2048
        roof = kernel_roofline("harvest", "hv_band_fft_kernel",
                               "FP64-FFT bound: %d channels x one inverse real FFT-2048 per block; fp64 figures beside it; "
                               "the whole path moves 400 B/frame (SURVEY.md 8d)" % nch,
                               kms, steps, frames, dec_per_frame * 8, fs, flops_per_launch=flops)
        roof["pipeline"] = {"bytes_per_frame": 400, "achieved_gbs": round(value / env.world * 400 / 1e9, 3),
                            "frac_hbm": round(value / env.world * 400 / 1e9 / HBM_PEAK_GBS, 7),
                            "achieved_fp64_tflops": round(value / env.world * 0.39e6 / 1e12, 3),
                            "frac_fp64": round(value / env.world * 0.39e6 / 1e12 / FP64_PEAK_TFLOPS, 4),
                            "note": "0.39 MFLOP/frame is the reference's FFT count (SURVEY.md 8d); the kernels here "
                                    "execute fewer (block convolution, Goertzel sums at six bins)"}
        line = side_line(env, "Harvest @48kHz, 1ms hop",
                         {"workload": "configs[2]: %d synthetic utterances (%g-%g s) per GPU, fs %d, hop %g ms, Harvest "
                                      "(71-800 Hz), f0 resident in HBM" % (len(xs), env.args.dur[0], env.args.dur[1], fs, fp)},
                         value, steps, warmup, elapsed, frames, roof, ramp)
        if env.world == 1 and cpu:
            from oracle.bindings import Oracle, Reference
            lib = Reference() if Reference.available() else Oracle()
            got = batch.harvest(x)[1].cpu().numpy()
            nfr, tcpu, df0, n = 0, 0.0, 0.0, 0
            for u in range(min(len(xs), 6)):
                a = time.perf_counter()
                tc, fc = lib.harvest(xs[u], fs, fp)
                tcpu += time.perf_counter() - a
                nfr += len(fc)
                n += 1
                df0 = max(df0, float(np.abs(got[batch.frame_offsets[u]:batch.frame_offsets[u + 1]] - fc).max()))
                if tcpu > 10:
                    break
            line["cpu_baseline"] = {"value": round(nfr / tcpu, 1), "unit": "frames/s", "cores": 1, "kind": lib.kind,
                                    "sample": "first %d utterances of the same batch (%d frames), Harvest, single thread, "
                                              "%.1f s" % (n, nfr, tcpu)}
            line["parity"] = {"vs": lib.kind, "utterances": n, "max_abs_dF0_hz": df0}
    batch.close()
    return line


def synthesis_bench(env, ctx, xs, feats, fs, fp, steps, warmup, prewarm, cpu=True):
    """configs[4]: Synthesis only from precomputed f0 / sp / ap, 1024 feature sets per GPU.  `feats` (f0, sp, ap,
    frame counts, output lengths of 256 analysed utterances) makes the 1024 sets from those 256 four times over (the
    compact side run of the default invocation); without it the sets are the analyses of `xs`."""
    torch, W = env.torch, env.W
    if feats is None:
        ba = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(v) for v in xs])
        t, f0, sp, ap = ba.analyze(torch.from_numpy(np.concatenate(xs)).cuda())
        T, Y = np.diff(ba.frame_offsets).tolist(), np.diff(ba.out_offsets).tolist()
        ba.close()
        how = "%d synthetic utterances (%g-%g s) analysed beforehand" % (len(xs), env.args.dur[0], env.args.dur[1])
        src = xs
    else:
        f0a, spa, apa, Ta, Ya = feats
        rep = max(1, 1024 // len(Ta))
        f0, sp, ap = f0a.repeat(rep), spa.repeat(rep, 1), apa.repeat(rep, 1)
        T, Y = Ta * rep, Ya * rep
        how = "%d feature sets = the headline's %d analysed utterances %d times over" % (len(T), len(Ta), rep)
        src = xs
    batch = W.WorldBatch(ctx, W.default_params(fs, fp), f0_lengths=T, y_lengths=Y)
    frames = int(batch.total_frames)
    y = torch.empty(int(batch.total_out), dtype=torch.float64, device="cuda")
    step = lambda: batch.synthesize(f0, sp, ap, out=y)
    ctx.timing_enable(False)
    ramp = clock_ramp(step, env.barrier, prewarm, steps) if prewarm > 0 else None
    for _ in range(warmup):
        step()
    env.barrier()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    env.barrier()
    elapsed = time.perf_counter() - t0
    kms = {k: ctx.timing_query(k) for k in SYNTHESIS_KERNELS}
    ctx.timing_enable(False)
    elapsed, total = env.reduce(elapsed, frames)
    line = None
    if env.rank == 0:
        value = total * steps / elapsed
        bm = byte_model(fs, fp, batch.fft_size)
        F = batch.fft_size
        # per pulse: minimum-phase response (3 real + 1 complex transform of F points on F/2-point engines) plus the
        # aperiodic part; counted as 7 (voiced) / 4 (unvoiced) real FFTs of F points, about 5.8 per frame
        flops = frames * 5.8 * 2.5 * F * math.log2(F)
        roof = kernel_roofline("synthesis", "synth_pulse_kernel",
                               "FP64-FFT bound (7 / 4 transforms per voiced / unvoiced pulse); the kernel must read f0 and one "
                               "sp and ap row per frame, its per-pulse responses are intermediates",
                               kms, steps, frames, bm["pulse"], fs, flops_per_launch=flops)
        roof["pipeline"] = {"bytes_per_frame": bm["synthesis"],
                            "achieved_gbs": round(value / env.world * bm["synthesis"] / 1e9, 3),
                            "frac_hbm": round(value / env.world * bm["synthesis"] / 1e9 / HBM_PEAK_GBS, 6)}
        line = side_line(env, "Synthesis-only @16kHz, 5ms hop",
                         {"workload": "configs[4]: %s, Synthesis only (fft %d), fp64, y resident in HBM" % (how, F)},
                         value, steps, warmup, elapsed, frames, roof, ramp)
        if env.world == 1 and cpu:
            from oracle.bindings import Oracle, Reference
            lib = Reference() if Reference.available() else Oracle()
            fo, yo = batch.frame_offsets, batch.out_offsets
            yh = y.cpu().numpy()
            nfr, tcpu, dy, n = 0, 0.0, 0.0, 0
            for u in range(min(batch.n_utt, 24)):
                g = slice(fo[u], fo[u + 1])
                a_f0, a_sp, a_ap = f0[g].cpu().numpy(), sp[g].cpu().numpy(), ap[g].cpu().numpy()
                a = time.perf_counter()
                yy = lib.synthesis(a_f0, a_sp, a_ap, F, fp, fs, int(yo[u + 1] - yo[u]))
                tcpu += time.perf_counter() - a
                nfr += fo[u + 1] - fo[u]
                n += 1
                dy = max(dy, float(np.abs(yh[yo[u]:yo[u + 1]] - yy).max()))
                if tcpu > 8:
                    break
            line["cpu_baseline"] = {"value": round(nfr / tcpu, 1), "unit": "frames/s", "cores": 1, "kind": lib.kind,
                                    "sample": "the first %d feature sets of the same batch (%d frames), Synthesis, single "
                                              "thread, %.1f s" % (n, nfr, tcpu)}
            line["parity"] = {"vs": lib.kind, "utterances": n, "max_abs_dy": dy}
    batch.close()
    return line


def codec_bench(env, ctx, xs, fs, fp, steps, warmup, prewarm, cpu=True):
    """the recipe's coded features from resident sp/ap, and the decoders back."""
    torch, W = env.torch, env.W
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    batch = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(v) for v in xs])
    frames = int(batch.total_frames)
    t, f0, sp, ap = batch.analyze(x)
    names = ("codec_code_sp_kernel", "codec_decode_sp_kernel", "codec_code_ap_kernel", "codec_decode_ap_kernel")

    def step():
        batch.recipe_features(f0, sp, ap, 50, 25)
        csp = batch.code_spectral_envelope(sp, 50)
        cap = batch.code_aperiodicity(ap)
        batch.decode_spectral_envelope(csp)
        batch.decode_aperiodicity(cap)

    ctx.timing_enable(False)
    ramp = clock_ramp(step, env.barrier, prewarm, steps) if prewarm > 0 else None
    for _ in range(warmup):
        step()
    env.barrier()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    env.barrier()
    elapsed = time.perf_counter() - t0
    kms = {k: ctx.timing_query(k)[0] / steps for k in names}
    # the synth CLI's way back (WorldMi355RecipeDecode), outside the timed step: reported beside it
    lf0, mgc, bap = batch.recipe_features(f0, sp, ap, 50, 25)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        batch.recipe_decode(lf0, mgc, bap)
    torch.cuda.synchronize()
    kms["recipe_decode_call_ms (not in value)"] = (time.perf_counter() - t1) * 1e3 / steps
    kms["codec_bap_decode_kernel (not in value)"] = ctx.timing_query("codec_bap_decode_kernel")[0] / steps
    ctx.timing_enable(False)
    elapsed, total = env.reduce(elapsed, frames)
    line = None
    if env.rank == 0:
        value = total * steps / elapsed
        # per frame: recipe packing reads sp+ap (2 x 4104 B), coders read them again, decoders write them
        bpf = 4 * 4104 + 2 * 4104
        roof = {"bound": "hbm", "achieved": round(value / env.world * bpf / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(value / env.world * bpf / 1e9 / HBM_PEAK_GBS, 7), "traffic": None, "bytes_per_unit": bpf,
                "kernel_ms_per_step": {k: round(v, 4) for k, v in kms.items()}}
        line = side_line(env, "feature codec (recipe packing + code + decode) @16kHz",
                         {"workload": "configs[1] features: %d synthetic utterances (%g-%g s) per GPU, recipe packing + "
                                      "CodeSpectralEnvelope + CodeAperiodicity + both decoders"
                                      % (len(xs), env.args.dur[0], env.args.dur[1])},
                         value, steps, warmup, elapsed, frames, roof, ramp)
        if env.world == 1 and cpu:
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
    batch.close()
    return line


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


def cpu_baseline_o2(xs, fs, fp, F, n_cpu, budget_s=10.0):
    """The reference's sources compiled at -O2 (oracle/_ref/libworld_ref_O2.so; SURVEY.md 8(d) asks for the CPU
    baseline at the shipped -O1 and at -O2), single thread, on the first utterances of the same batch.  Timing only:
    parity is taken against the shipped flags."""
    from oracle.bindings import Reference
    if not Reference.available(o2=True):
        return None
    lib = Reference(o2=True)
    frames, cpu_time, n = 0, 0.0, 0
    for x in xs[:n_cpu]:
        a = time.perf_counter()
        t, f0 = lib.dio(x, fs, fp)
        f0 = lib.stonemask(x, fs, t, f0)
        sp = lib.cheaptrick(x, fs, t, f0, -0.15, F)
        ap = lib.d4c(x, fs, t, f0, F, 0.0)
        lib.synthesis(f0, sp, ap, F, fp, fs)
        cpu_time += time.perf_counter() - a
        frames += len(f0)
        n += 1
        if cpu_time > budget_s:
            break
    return {"value": round(frames / cpu_time, 1), "unit": "frames/s", "cores": 1, "kind": "reference",
            "flags": "-O2 (the reference ships -O1, externs/WORLD_v2/makefile:5: that build is cpu_baseline)",
            "sample": "first %d utterances of the same batch (%d frames), Dio+StoneMask+CheapTrick+D4C+Synthesis, "
                      "single thread, %.1f s" % (n, frames, cpu_time)}


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
        if time.perf_counter() - t0 > 25.0:
            n_cpu = u + 1
            break
    cpu = {"value": round(frames / cpu_time, 1), "unit": "frames/s", "cores": 1, "kind": lib.kind,
           "flags": "-O1 -g (the reference's own, externs/WORLD_v2/makefile:5)" if lib.kind == "reference" else "-O2 (oracle/Makefile)",
           "sample": "first %d utterances of the same batch (%d frames), Dio+StoneMask+CheapTrick+D4C+Synthesis, "
                     "single thread, %.1f s" % (n_cpu, frames, cpu_time)}
    parity = {"vs": lib.kind, "utterances": n_cpu, "max_abs_dF0_hz": df0, "sp_rmse": (se_sp / cnt) ** 0.5,
              "ap_rmse": (se_ap / cnt) ** 0.5, "max_abs_dy": dy}
    return cpu, parity


if __name__ == "__main__":
    main()
