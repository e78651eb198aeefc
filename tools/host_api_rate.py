"""PCIe-inclusive rates (never bench.py's `value`).

  a) the drop-in C ABI with host pointers, one utterance per call like the reference CLI
     (Dio -> StoneMask -> CheapTrick -> D4C -> Synthesis; every call stages its arguments over PCIe);
  b) the batched API with the waveforms starting in host memory and f0/sp/ap/y ending there.

Run on the GPU box: python tools/host_api_rate.py [--utts 32]"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap_ = argparse.ArgumentParser()
ap_.add_argument("--utts", type=int, default=32)
args = ap_.parse_args()
pkg = importlib.import_module("hts-train-world_amd")
W, sd, capi = pkg.world, pkg.synth_data, importlib.import_module("hts-train-world_amd.capi")
fs, fp = 16000, 5.0
xs = sd.make_batch(256, fs, (2.0, 8.0), first=0, workers=8)


def one(x):
    t, f0 = capi.dio(x, fs, fp)
    f0 = capi.stonemask(x, fs, t, f0)
    F = capi.cheaptrick_fft_size(fs)
    sp = capi.cheaptrick(x, fs, t, f0)
    ap = capi.d4c(x, fs, t, f0, F, 0.0)
    y = capi.synthesis(f0, sp, ap, F, fp, fs)
    return len(f0), y


one(max(xs[:args.utts], key=len))               # context creation, first-use allocations, randn table at full size
t0 = time.perf_counter()
frames = sum(one(x)[0] for x in xs[:args.utts])
dt = time.perf_counter() - t0
print("a) C ABI, host pointers, per utterance: %d utterances, %d frames, %.1f ms/utterance, %.0f frames/s"
      % (args.utts, frames, dt / args.utts * 1e3, frames / dt))

ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(v) for v in xs])
hx = torch.from_numpy(np.concatenate(xs)).pin_memory()


def batched():
    x = hx.cuda(non_blocking=True)
    t, f0, sp, ap = b.analyze(x)
    y = b.synthesize(f0, sp, ap)
    out = [v.cpu() for v in (f0, sp, ap, y)]
    torch.cuda.synchronize()
    return out


batched()
t0 = time.perf_counter()
for _ in range(3):
    batched()
dt = (time.perf_counter() - t0) / 3
print("b) batched, host -> device -> host (pageable results): %d frames, %.1f ms/step, %.0f frames/s"
      % (b.total_frames, dt * 1e3, b.total_frames / dt))
outs = None
pinned = [torch.empty(v.shape, dtype=v.dtype).pin_memory() for v in batched()]


def batched_pinned():
    x = hx.cuda(non_blocking=True)
    t, f0, sp, ap = b.analyze(x)
    y = b.synthesize(f0, sp, ap)
    for dst, src in zip(pinned, (f0, sp, ap, y)):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()


batched_pinned()
t0 = time.perf_counter()
for _ in range(3):
    batched_pinned()
dt = (time.perf_counter() - t0) / 3
print("c) batched, pinned host buffers both ways: %.1f ms/step, %.0f frames/s" % (dt * 1e3, b.total_frames / dt))
