"""Throughput of the recipe's real setting (48 kHz, 5 ms, fft 2048) on 64 synthetic utterances, stage by stage.
Run on the GPU box: python tools/rate_48k.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("hts-train-world_amd")
W, sd = pkg.world, pkg.synth_data
fs, fp = 48000, 5.0
xs = sd.make_batch(64, fs, (2.0, 8.0), first=0, workers=8)
x = torch.from_numpy(np.concatenate(xs)).cuda()
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(v) for v in xs])
def tm(f, *a, n=3):
    f(*a); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f(*a)
    torch.cuda.synchronize(); return r, (time.perf_counter() - t0) / n * 1e3
(t, f0), t_dio = tm(b.dio, x)
f0, t_sm = tm(b.stonemask, x, t, f0)
sp, t_ct = tm(b.cheaptrick, x, t, f0)
ap, t_d4 = tm(b.d4c, x, t, f0)
y, t_sy = tm(b.synthesize, f0, sp, ap)
(_, _, _, _, _), t_all = tm(b.analyze_synthesize, x)
(lf0, mgc, bap), t_rf = tm(b.recipe_features, f0, sp, ap, 50, 25)
ctx.timing_enable(True)
b.analyze_synthesize(x); b.analyze_synthesize(x)
ks = ("dio_lowcut_kernel", "dio_band_kernel", "dio_candidate_kernel", "dio_fix_kernel", "stonemask_kernel", "cheaptrick_kernel",
      "d4c_lovetrain_kernel", "d4c_kernel", "synth_inc_kernel", "synth_timebase_kernel", "synth_search_kernel",
      "synth_pulse_kernel", "synth_ola_kernel")
print({k: round(ctx.timing_query(k)[0] / 2, 3) for k in ks})
print("frames %d fft %d: dio %.2f stonemask %.2f cheaptrick %.2f d4c %.2f synthesis %.2f | one call %.2f ms -> %.2f M frames/s | recipe features %.2f ms"
      % (b.total_frames, b.fft_size, t_dio, t_sm, t_ct, t_d4, t_sy, t_all, b.total_frames / t_all / 1e3, t_rf))
