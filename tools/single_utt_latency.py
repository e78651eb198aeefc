"""Latency of the device-resident API on ONE utterance (where a batch of one spends its time).
Run on the GPU box: python tools/single_utt_latency.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("hts-train-world_amd")
W, sd = pkg.world, pkg.synth_data
fs, fp = 16000, 5.0
x_h = sd.make_utterance(5, fs, duration=5.0)
x = torch.from_numpy(x_h).cuda()
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
def tm(f, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3
for rnd in range(3):
    b, t_new = tm(W.WorldBatch, ctx, W.default_params(fs, fp), x_lengths=[len(x_h)])
    (t, f0), t_dio = tm(b.dio, x)
    f0, t_sm = tm(b.stonemask, x, t, f0)
    sp, t_ct = tm(b.cheaptrick, x, t, f0)
    ap, t_d4c = tm(b.d4c, x, t, f0)
    ap, t_d4c2 = tm(b.d4c, x, t, f0)
    y, t_sy = tm(b.synthesize, f0, sp, ap)
    y, t_sy2 = tm(b.synthesize, f0, sp, ap)
    _, t_close = tm(b.close)
    print("round %d frames %d: new %.2f dio %.2f stonemask %.2f cheaptrick %.2f d4c %.2f (again %.2f) synthesis %.2f (again %.2f) close %.2f ms"
          % (rnd, len(f0), t_new, t_dio, t_sm, t_ct, t_d4c, t_d4c2, t_sy, t_sy2, t_close))
