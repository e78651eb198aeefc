"""Load balance of round-robin frame dealing: active (voiced, LoveTrain-passed) frames per persistent wave.
Run on the GPU box: python tools/deal_stats.py"""
import importlib
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

pkg = importlib.import_module("hts-train-world_amd")
W, sd = pkg.world, pkg.synth_data
fs, fp = 16000, 5.0
xs = sd.make_batch(256, fs, (2.0, 8.0), first=0, workers=8)
x = torch.from_numpy(np.concatenate(xs)).cuda()
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(v) for v in xs])
t, f0, sp, ap = b.analyze(x)
torch.cuda.synchronize()
act = ((f0 > 0) & (ap[:, 0] < 0.99)).cpu().numpy()
print("frames", len(act), "f0>0", int((f0 > 0).sum()), "active", int(act.sum()))
for G in (2048, 3072):
    n = (len(act) + G - 1) // G * G
    a = np.zeros(n, bool); a[:len(act)] = act
    per = a.reshape(-1, G).sum(0)
    print(G, "per-wave active: mean %.1f max %d min %d -> max/mean %.3f" % (per.mean(), per.max(), per.min(), per.max() / per.mean()))
