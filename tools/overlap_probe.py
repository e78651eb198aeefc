"""Do device-to-host copies overlap the analysis kernels?  Times the kernels alone, a pinned D2H of the float32
features alone, and both issued together on different streams.  Run on the GPU box: python tools/overlap_probe.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("hts-train-world_amd")
W, sd = pkg.world, pkg.synth_data
fs, fp = 16000, 5.0
xs = sd.make_batch(256, fs, (2.0, 8.0), first=0, workers=8)
own = os.environ.get("OWN_STREAM", "0") == "1"
main = torch.cuda.Stream() if own else torch.cuda.current_stream()
with torch.cuda.stream(main):
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    ctx = W.Context(stream_ptr=main.cuda_stream)
    b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(v) for v in xs])
    out = b.analyze_synthesize(x)
    dev = torch.empty(int(b.total_frames), 2 * b.bins + 1, dtype=torch.float32, device="cuda")
host = torch.empty(dev.shape, dtype=torch.float32, pin_memory=True)
down = torch.cuda.Stream()
torch.cuda.synchronize()


def kernels():
    with torch.cuda.stream(main):
        b.analyze_synthesize(x, out=out[:4], y=out[4])


def copy():
    with torch.cuda.stream(down):
        host.copy_(dev, non_blocking=True)


def tm(*fs_, n=4):
    for f in fs_:
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fs_:
            f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("main stream: %s (ptr %d)" % ("own non-default stream" if own else "torch default stream", main.cuda_stream))
print("kernels alone %.2f ms | D2H of %.2f GB alone %.2f ms | both %.2f ms (copy first) | both %.2f ms (kernels first)"
      % (tm(kernels), dev.numel() * 4 / 1e9, tm(copy), tm(copy, kernels), tm(kernels, copy)))
