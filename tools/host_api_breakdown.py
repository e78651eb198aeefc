import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("hts-train-world_amd")
sd, capi = pkg.synth_data, importlib.import_module("hts-train-world_amd.capi")
fs, fp = 16000, 5.0
xs = sd.make_batch(16, fs, (2.0, 8.0), first=0, workers=8)
acc = {}
def tm(name, f, *a):
    t0 = time.perf_counter(); r = f(*a); acc[name] = acc.get(name, 0) + time.perf_counter() - t0; return r
def one(x):
    t, f0 = tm("dio", capi.dio, x, fs, fp)
    f0 = tm("stonemask", capi.stonemask, x, fs, t, f0)
    F = capi.cheaptrick_fft_size(fs)
    sp = tm("cheaptrick", capi.cheaptrick, x, fs, t, f0)
    ap = tm("d4c", capi.d4c, x, fs, t, f0, F, 0.0)
    y = tm("synthesis", capi.synthesis, f0, sp, ap, F, fp, fs)
one(max(xs, key=len)); acc.clear()
for x in xs: one(x)
print({k: round(v / len(xs) * 1e3, 2) for k, v in acc.items()})
