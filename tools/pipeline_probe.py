"""Timeline of hts-train-world_amd/pipeline.py: per step, when its kernels ended and when its download ended
(HIP events, ms since the first step's start).  Run on the GPU box: python tools/pipeline_probe.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("hts-train-world_amd")
W, sd, pl = pkg.world, pkg.synth_data, pkg.pipeline
fs, fp = 16000, 5.0
xs = sd.make_batch(256, fs, (2.0, 8.0), first=0, workers=8)
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
pipe = pl.HostPipeline(ctx, W.default_params(fs, fp), [len(x) for x in xs], synthesis=True)
x16 = pl.to_int16(np.concatenate(xs))
for xb in pipe.x_pinned:
    xb.numpy()[:] = x16
for _ in range(2):
    pipe.result(pipe.submit())
K = 6
mk = lambda: torch.cuda.Event(enable_timing=True)
start = mk(); start.record(pipe.compute)
marks, host = [], []
prev = None
t0 = time.perf_counter()
pipe.feed()
for k in range(K):
    h0 = time.perf_counter() - t0
    if k + 1 < K:
        pipe.feed()
    slot = pipe.submit()
    a, b_ = mk(), mk()
    a.record(pipe.compute); b_.record(pipe.down)
    marks.append((a, b_))
    h1 = time.perf_counter() - t0
    if prev is not None:
        pipe.result(prev)
    h2 = time.perf_counter() - t0
    host.append((h0 * 1e3, h1 * 1e3, h2 * 1e3))
    prev = slot
pipe.result(prev)
torch.cuda.synchronize()
for k, ((a, b_), h) in enumerate(zip(marks, host)):
    print("step %d: host submit %.1f -> %.1f, result(prev) returned %.1f | kernels+conversions done %.1f, download done %.1f"
          % (k, h[0], h[1], h[2], start.elapsed_time(a), start.elapsed_time(b_)))
print("total %.1f ms for %d steps" % ((time.perf_counter() - t0) * 1e3, K))
