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
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector peak
# Algorithmic bytes per frame, fp64 C-API layout, 16 kHz / 5 ms / fft 1024:
#   analysis  : 80 samples*8 + t 8 + f0 8 + sp 513*8 + ap 513*8 = 8864
#   synthesis : f0 8 + sp 4104 + ap 4104 + 80*8                 = 8856
BYTES_PER_FRAME = 8864 + 8856
FLOPS_PER_FRAME = 0.7e6
# d4c_kernel alone: reads its 80 new samples + t + f0 + ap0 (664 B), writes one ap row (4104 B)
D4C_BYTES_PER_FRAME = 80 * 8 + 8 + 8 + 8 + 513 * 8
D4C_FLOPS_PER_VOICED_FRAME = 6 * 2.5 * 2048 * 11   # 6 real FFTs of 2048 (+ scans etc., not counted)
# HBM-side bytes of d4c_kernel from the PMC counters, separate --pmc passes (tools/pmc_hbm.sh) on 64 utterances
# (67 884 frames): FETCH_SIZE 825 078 KB, WRITE_SIZE 1 239 844 KB (profiles/r01_f_pmc_{fetch,write}_size_64utts.csv).
# WRITE_SIZE is exact for this access width (cheaptrick_kernel in the same pass: 4104 B/frame = 513 doubles);
# FETCH_SIZE is taken as reported (8 B/lane reads; the guide's x2 correction is for 16 B/lane streaming reads).
# 14.6 KB/frame of the writes and most of the reads are the kernel's 79 spilled VGPRs going through scratch.
D4C_HBM_BYTES_PER_FRAME = (825078.0 + 1239844.0) * 1024 / 67884


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--dur", type=float, nargs=2, default=(2.0, 8.0))
    ap.add_argument("--gather", action="store_true", help="include the RCCL feature gather in the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate-calls", action="store_true",
                    help="WorldMi355Analyze then WorldMi355Synthesis instead of WorldMi355AnalyzeSynthesize")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl (= RCCL; the driver's launch) or gloo: rehearsal of the N > 1 control path with several "
                         "ranks sharing one GPU, where RCCL refuses duplicate devices")
    ap.add_argument("--cpu-utts", type=int, default=48, help="utterances of the CPU baseline sample")
    ap.add_argument("--workers", type=int, default=0, help="processes for synthetic data generation (0 = auto)")
    ap.add_argument("--workload", choices=["analysis_synthesis", "harvest", "synthesis", "codec"], default="analysis_synthesis",
                    help="analysis_synthesis = configs[1] (the headline metric); harvest = configs[2] (48 kHz, 1 ms, "
                         "64 utterances); synthesis = configs[4] (Synthesis only from precomputed features); codec = the recipe's "
                         "coded lf0/mgc/bap from resident features plus the decoders (SURVEY.md 8(f))")
    return ap.parse_args()


def main():
    args = parse()
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
    xs = sd.make_batch(utts, fs, tuple(args.dur), first=rank * utts, workers=workers)
    # the CPU path on all host cores of this process' share, also before the GPU is touched
    cpu_all = None
    if (world == 1 and workers > 1 and not args.no_cpu_baseline and args.workload == "analysis_synthesis"):
        cpu_all = cpu_all_cores(xs, fs, fp, workers, 6 * workers)

    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if args.backend == "gloo":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

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

    def step():
        if args.separate_calls:
            t, f0, sp, ap = batch.analyze(x, out=outs)
            batch.synthesize(f0, sp, ap, out=y)
        else:
            # the same launches as one call: Synthesis' f0-only first part runs beside CheapTrick / D4C
            t, f0, sp, ap, _ = batch.analyze_synthesize(x, out=outs, y=y)
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

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = {}
    for k in ("dio_lowcut_kernel", "dio_band_kernel", "dio_candidate_kernel", "dio_fix_kernel", "stonemask_kernel",
              "cheaptrick_kernel", "d4c_lovetrain_kernel", "d4c_kernel", "synth_inc_kernel",
              "synth_timebase_kernel", "synth_search_kernel", "synth_pulse_kernel", "synth_ola_kernel"):
        ms, n = ctx.timing_query(k)
        kernel_ms[k] = (ms, n)
    ctx.timing_enable(False)

    tt = torch.tensor([elapsed, float(frames)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, total_frames = float(tmax[0]), float(tsum[1])
    else:
        elapsed_max, total_frames = elapsed, float(frames)

    if rank == 0:
        value = total_frames * args.steps / elapsed_max
        d4c_ms, d4c_n = kernel_ms["d4c_kernel"]
        d4c_avg_s = d4c_ms / max(1, d4c_n) * 1e-3
        voiced = int((outs[1] > 0).sum().item())
        roof = {
            "bound": "hbm", "kernel": "d4c_kernel",
            "achieved": round(frames * D4C_BYTES_PER_FRAME / d4c_avg_s / 1e9, 3) if d4c_avg_s > 0 else None,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(frames * D4C_BYTES_PER_FRAME / d4c_avg_s / 1e9 / HBM_PEAK_GBS, 6) if d4c_avg_s > 0 else None,
            "traffic": round(frames * D4C_HBM_BYTES_PER_FRAME),
            "traffic_note": "HBM-side read + write bytes per launch from PMC FETCH_SIZE / WRITE_SIZE (separate passes, "
                            "profiles/r01_f_pmc_*_size_64utts.csv, per-frame figure x frames); above the algorithmic "
                            "bytes because 79 spilled VGPRs travel through scratch",
            "launch_ms": round(d4c_avg_s * 1e3, 4), "units_per_launch": frames,
            "bytes_per_unit": D4C_BYTES_PER_FRAME,
            "note": "FP64-FFT/LDS bound, not HBM bound (SURVEY.md 8d); fp64 figures beside it",
            "fp64": {"achieved_tflops": round(voiced * D4C_FLOPS_PER_VOICED_FRAME / d4c_avg_s / 1e12, 3) if d4c_avg_s > 0 else None,
                     "peak_tflops": FP64_PEAK_TFLOPS,
                     "frac": round(voiced * D4C_FLOPS_PER_VOICED_FRAME / d4c_avg_s / 1e12 / FP64_PEAK_TFLOPS, 5) if d4c_avg_s > 0 else None},
            "pipeline": {"achieved_gbs": round(value / world * BYTES_PER_FRAME / 1e9, 3),
                         "frac_hbm": round(value / world * BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 6),
                         "achieved_fp64_tflops": round(value / world * FLOPS_PER_FRAME / 1e12, 4)},
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in kernel_ms.items()},
        }
        cpu = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, parity = cpu_baseline_and_parity(xs, fs, fp, batch, outs, y, args.cpu_utts)
        line = {
            "metric": "WORLD analysis+synthesis frames/sec @16kHz, 5ms hop",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: batch of %d synthetic 16 kHz utterances (%g-%g s) per GPU, "
                                   "Dio+StoneMask+CheapTrick+D4C then Synthesis (%s), fp64, features resident in HBM"
                                   % (args.utts, args.dur[0], args.dur[1],
                                      "two calls" if args.separate_calls else
                                      "one call: Synthesis' f0-only part on a second stream beside CheapTrick/D4C"),
                       "fs": fs, "frame_period_ms": fp, "fft_size": batch.fft_size,
                       "utterances_per_gpu": args.utts, "frames_per_gpu": frames,
                       "parallelism": "utterance-sharded x%d%s" % (world, "+gather" if args.gather else "")},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if cpu_all:
            line["cpu_baseline_all_cores"] = cpu_all
        if parity:
            line["parity"] = parity
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def workload_spec(args):
    """(fs, frame period, utterances per GPU) of the selected workload."""
    if args.workload == "harvest":
        return 48000, 1.0, (args.utts if args.utts != 256 else 64)
    if args.workload == "synthesis":
        return 16000, 5.0, (args.utts if args.utts != 256 else 1024)
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
