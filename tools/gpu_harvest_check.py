import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("hts-train-world_amd")
from oracle.bindings import Oracle
W, sd = pkg.world, pkg.synth_data
ctx = W.Context()
o = Oracle()
for fs, fp, idxs, dur in [(16000, 5.0, [3, 6], (1.0, 2.0)), (48000, 1.0, [4, 8, 9], (1.0, 2.5)), (22050, 5.0, [7], (1.5, 1.5))]:
    xs = [sd.make_utterance(i, fs, dur) for i in idxs]
    p = W.default_params(fs, fp)
    b = W.WorldBatch(ctx, p, x_lengths=[len(x) for x in xs])
    xc = torch.from_numpy(np.concatenate(xs)).cuda()
    t0 = time.time(); t, f0 = b.harvest(xc); torch.cuda.synchronize(); t1 = time.time()
    t, f0 = b.harvest(xc); torch.cuda.synchronize(); t2 = time.time()
    f0 = f0.cpu().numpy(); t = t.cpu().numpy()
    ro = [o.harvest(x, fs, fp) for x in xs]
    fo = np.concatenate([r[1] for r in ro]); to = np.concatenate([r[0] for r in ro])
    d = np.abs(f0 - fo)
    print(f"fs={fs} fp={fp}: frames {len(fo)} voiced {(fo>0).sum()} vuv mismatch {((f0>0)!=(fo>0)).sum()} max|d| {d.max():.3e} nbad(>1e-6) {(d>1e-6).sum()} t ok {np.array_equal(t,to)}  first {1e3*(t1-t0):.1f} ms second {1e3*(t2-t1):.1f} ms", flush=True)
    if (d > 1e-6).any():
        bad = np.where(d > 1e-6)[0]
        print("  bad idx", bad[:20], f0[bad[:8]], fo[bad[:8]])
    b.close()
