import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("hts-train-world_amd")
from oracle.bindings import Oracle
W, sd = pkg.world, pkg.synth_data
fs = 16000
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
o = Oracle()
x = sd.make_utterance(1, fs, (2.0, 4.0))
t, f0 = o.dio(x, fs); ref = o.stonemask(x, fs, t, f0)
p = W.default_params(fs, 5.0)
b = W.WorldBatch(ctx, p, x_lengths=[len(x)])
xc = torch.from_numpy(x).cuda()
out = b.stonemask(xc, torch.from_numpy(t).cuda(), torch.from_numpy(f0).cuda()).cpu().numpy()
d = np.abs(out - ref)
bad = np.where(d > 1e-9)[0]
print("bad frames", bad, d[bad], out[bad], ref[bad], f0[bad])
# perturb f0 slightly around frame 282 and see
f0b = f0.copy()
for eps in (0.0, 1e-9, 1e-6, 1e-3):
    f0b[282] = f0[282] + eps
    r2 = o.stonemask(x, fs, t, f0b)[282]
    g2 = b.stonemask(xc, torch.from_numpy(t).cuda(), torch.from_numpy(f0b).cuda()).cpu().numpy()[282]
    print(eps, r2, g2, g2 - r2)
