import sys, time, importlib
sys.path.insert(0, '/root/repo')
import numpy as np, torch
pkg = importlib.import_module('hts-train-world_amd')
W, sd = pkg.world, pkg.synth_data
fs, fp = 16000, 5.0
xs = sd.make_batch(256, fs, (2.0, 8.0), workers=8)
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(x) for x in xs])
x = torch.from_numpy(np.concatenate(xs)).cuda()
ts = []
for i in range(700):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.analyze_synthesize(x)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
for k in range(0, 700, 50):
    print(k, 'mean %.2f min %.2f max %.2f' % (ts[k:k+50].mean(), ts[k:k+50].min(), ts[k:k+50].max()))
