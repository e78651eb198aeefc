import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
pkg = importlib.import_module("hts-train-world_amd")
from oracle.bindings import Oracle, Reference
W, sd = pkg.world, pkg.synth_data
o = Reference() if Reference.available() else Oracle()
print("checker", o.kind)
fs = 16000
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
def chain(x):
    t, f0 = o.dio(x, fs); f0 = o.stonemask(x, fs, t, f0)
    sp = o.cheaptrick(x, fs, t, f0); ap = o.d4c(x, fs, t, f0, 1024, 0.0)
    y = o.synthesis(f0, sp, ap, 1024, 5.0, fs)
    return f0, sp, ap, y
for name, x in (("zeros", np.zeros(8000)), ("tiny", sd.make_utterance(3, fs, duration=0.5) * 1e-6),
                ("long40s", sd.make_utterance(5, fs, duration=40.0)), ("clipped", np.clip(sd.make_utterance(6, fs, duration=1.0) * 10, -1, 1))):
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    t, f0, sp, ap = b.analyze(torch.from_numpy(x).cuda()); y = b.synthesize(f0, sp, ap)
    torch.cuda.synchronize()
    t0 = time.time(); r = chain(x); dt = time.time() - t0
    g = [v.cpu().numpy() for v in (f0, sp, ap, y)]
    print(name, "frames", len(r[0]), "cpu %.1fs" % dt, "dF0 %.2e" % np.abs(g[0] - r[0]).max(),
          "sp rel %.2e" % (np.abs(g[1] - r[1]) / np.abs(r[1])).max(), "ap %.2e" % np.abs(g[2] - r[2]).max(),
          "y %.2e" % np.abs(g[3] - r[3]).max(), "finite", all(np.isfinite(v).all() for v in g))
    b.close()
