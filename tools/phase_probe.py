"""Where a per-frame kernel's cycles go, phase by phase (debug build: make -C hts-train-world_amd/csrc EXTRA=-DWM_PHASE,
into a copy of the library named by WORLD_MI355_LIB).  Shader-clock totals over all waves of the analysis of a batch."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("hts-train-world_amd")
W, sd = pkg.world, pkg.synth_data
NAMES = {0: ["pipe", "centroids", "centroid dc", "hann frame", "power fft", "dc + smooth", "group delay", "band window",
             "band fft", "sort/peel/log", "row"],
         1: ["pipe", "frame", "power fft", "dc", "smooth", "noise + log", "fft 2", "lifter", "fft 3", "exp + row"],
         2: ["pipe", "spectra", "min phase 1", "mid", "min phase 2", "noise", "response"]}
unit = int(sys.argv[1]) if len(sys.argv) > 1 else 0
fs = 16000
xs = sd.make_batch(64, fs, (2.0, 8.0), workers=8)
lib = W.load_library()
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
x = torch.from_numpy(np.concatenate(xs)).cuda()
out = (C.c_ulonglong * 32)()
b.analyze_synthesize(x)
lib.WorldMi355DebugPhases(unit, out)
for _ in range(3):
    b.analyze_synthesize(x)
assert lib.WorldMi355DebugPhases(unit, out) == 0
tot = float(sum(out))
print("unit", unit, "total cycles %.3e" % tot)
for k, v in enumerate(out):
    if v:
        print("  %-14s %5.1f%%" % (NAMES[unit][k] if k < len(NAMES[unit]) else k, 100.0 * v / tot))
