"""Where the GPU's CheapTrick differs most from the compiled reference on the reference's own wav (real speech):
relative error per bin against the bin's level below the frame's strongest bin.  Needs oracle/_ref (vectors are
recomputed with the reference, full arrays)."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.bindings import Oracle, Reference  # noqa: E402

pkg = importlib.import_module("hts-train-world_amd")
W = pkg.world
for name in ("real_arctic_a0001", "real_vaiueo2d"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    x = g["x_i16"].astype(np.float64) / 32768.0
    fs, F = int(g["fs"]), int(g["fft_size"])
    ref = Reference() if Reference.available() else Oracle()
    sp_r = ref.cheaptrick(x, fs, g["t"], g["f0"], -0.15, F)
    ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    sp_g = b.cheaptrick(dev(x), dev(g["t"]), dev(g["f0"])).cpu().numpy()
    rel = np.abs(sp_g - sp_r) / sp_r
    level = sp_r / sp_r.max(axis=1, keepdims=True)
    print(name, ref.kind, "max rel", rel.max(), "max abs", np.abs(sp_g - sp_r).max())
    for lo, hi in ((1e-3, 2), (1e-6, 1e-3), (1e-9, 1e-6), (1e-12, 1e-9), (0, 1e-12)):
        m = (level >= lo) & (level < hi)
        if m.any():
            print("  level [%g, %g): bins %d  max rel %.2e  median rel %.2e" % (lo, hi, m.sum(), rel[m].max(), np.median(rel[m])))
    i, j = np.unravel_index(np.argmax(rel), rel.shape)
    print("  worst: frame", i, "bin", j, "f0", g["f0"][i], "value", sp_r[i, j], "row max", sp_r[i].max(), "row sum", sp_r[i].sum())
    o = Oracle()
    sp_o = o.cheaptrick(x, fs, g["t"], g["f0"], -0.15, F)
    print("  oracle vs", ref.kind, "max rel", (np.abs(sp_o - sp_r) / sp_r).max())
    b.close()
    ctx.close()
