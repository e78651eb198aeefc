"""The C-ABI shared library loads without a GPU and exports every entry point that include/*.h
declares (no compute calls here)."""
import ctypes
import glob
import os
import re
import subprocess

import pytest

from conftest import ROOT

DECL = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b([A-Z][A-Za-z0-9_]+)\s*\(", re.M)


def declared_symbols():
    names = set()
    for path in glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "include", "world", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
        for m in DECL.finditer(text):
            names.add(m.group(1))
    return names


def test_headers_declare_the_world_api():
    names = declared_symbols()
    for want in ("Dio", "InitializeDioOption", "GetSamplesForDIO", "Harvest", "InitializeHarvestOption",
                 "GetSamplesForHarvest", "StoneMask", "CheapTrick", "InitializeCheapTrickOption",
                 "GetFFTSizeForCheapTrick", "GetF0FloorForCheapTrick", "D4C", "InitializeD4COption", "Synthesis",
                 "GetNumberOfAperiodicities", "CodeAperiodicity", "DecodeAperiodicity", "CodeSpectralEnvelope",
                 "DecodeSpectralEnvelope", "WorldMi355CodeSpectralEnvelope", "WorldMi355RecipeFeatures", "WorldMi355RecipeDecode",
                 "WorldMi355ComposeCmp", "WorldMi355HtkHeader", "WorldMi355WriteFiles",
                 "WorldMi355Analyze", "WorldMi355AnalyzeSynthesize", "WorldMi355Synthesis", "WorldMi355CreateBatch"):
        assert want in names, want


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    missing = [n for n in sorted(declared_symbols()) if not hasattr(lib, n)]
    assert not missing, missing


def test_option_struct_layouts_match_reference(pkg):
    """dio.h:16-23, harvest.h:16-20, cheaptrick.h:16-20, d4c.h:16-18 (sizes on LP64)."""
    C = pkg.capi
    assert ctypes.sizeof(C.DioOption) == 48 and C.DioOption.speed.offset == 32 and C.DioOption.allowed_range.offset == 40
    assert ctypes.sizeof(C.HarvestOption) == 24
    assert ctypes.sizeof(C.CheapTrickOption) == 24 and C.CheapTrickOption.fft_size.offset == 16
    assert ctypes.sizeof(C.D4COption) == 8


def test_option_initialisers_and_size_helpers(pkg):
    """Pure host functions of the C ABI (no device needed): defaults of dio.cpp:649-665,
    cheaptrick.cpp:191-198/230-239, d4c.cpp:399-401, harvest.cpp:1257-1262."""
    C = pkg.capi
    L = C._lib()
    d = C.DioOption()
    L.InitializeDioOption(ctypes.byref(d))
    assert (d.f0_floor, d.f0_ceil, d.channels_in_octave, d.frame_period, d.speed, d.allowed_range) == \
        (71.0, 800.0, 2.0, 5.0, 1, 0.1)
    h = C.HarvestOption()
    L.InitializeHarvestOption(ctypes.byref(h))
    assert (h.f0_floor, h.f0_ceil, h.frame_period) == (71.0, 800.0, 5.0)
    c = C.CheapTrickOption()
    L.InitializeCheapTrickOption(16000, ctypes.byref(c))
    assert (c.q1, c.f0_floor, c.fft_size) == (-0.15, 71.0, 1024)
    L.InitializeCheapTrickOption(48000, ctypes.byref(c))
    assert c.fft_size == 2048
    assert abs(L.GetF0FloorForCheapTrick(16000, 1024) - 3.0 * 16000 / 1021.0) < 1e-12
    o = C.D4COption()
    L.InitializeD4COption(ctypes.byref(o))
    assert o.threshold == 0.85
    assert L.GetSamplesForDIO(16000, 53680, 5.0) == 672
    assert L.GetSamplesForHarvest(48000, 192000, 1.0) == 4001
    assert [L.GetNumberOfAperiodicities(fs) for fs in (16000, 22050, 44100, 48000)] == [1, 2, 5, 5]   # codec.cpp:212-215


def test_no_cpu_fallback_without_device(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        pkg.world.Context()


def test_product_library_does_not_link_the_oracle():
    so = os.path.join(ROOT, "hts-train-world_amd", "libworld_mi355.so")
    out = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert "orc_" not in out
    for src in glob.glob(os.path.join(ROOT, "hts-train-world_amd", "**", "*"), recursive=True):
        if src.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
            text = open(src).read()
            assert "oracle" not in text.replace("no oracle", ""), src


def test_library_exports_only_its_abi():
    """A library meant to be linked into someone else's binaries exports the WORLD entry points and the WorldMi355*
    extension, nothing else: no wm:: internals, no kernel handles, no standard-library template instances
    (csrc/exports.map + -fvisibility=hidden)."""
    so = os.path.join(ROOT, "hts-train-world_amd", "libworld_mi355.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    declared = declared_symbols()
    assert exported == {n for n in declared if not n.startswith("WM_")}, sorted(exported ^ declared)[:20]
    assert len([n for n in exported if not n.startswith("WorldMi355")]) == 19      # WORLD's own names


def test_drop_in_error_handler_instead_of_abort(pkg):
    """Without a handler a failing drop-in call aborts the process (the reference's functions return void); with
    WorldMi355SetErrorHandler the failure is reported and the call returns.  On a box without a GPU every drop-in call
    fails with WM_ERR_NO_DEVICE, which is what this exercises (in a child process: the abort path kills it)."""
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the call succeeds")
    code = r'''
import ctypes as C, importlib, sys
import numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module("hts-train-world_amd")
L = pkg.capi._lib()
seen = []
H = C.CFUNCTYPE(None, C.c_char_p, C.c_int, C.c_char_p, C.c_void_p)
def on_error(where, code, msg, user):
    seen.append((where.decode(), code, msg.decode()))
cb = H(on_error)
if sys.argv[1] == "handler":
    L.WorldMi355SetErrorHandler(cb, None)
x = np.zeros(1600)
t, f0 = pkg.capi.dio(x, 16000)
print("returned", seen)
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code, "handler"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    assert "returned [('CreateContext', 4," in r.stdout, r.stdout
    r = subprocess.run([sys.executable, "-c", code, "abort"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "libworld_mi355: CreateContext failed (code 4)" in r.stderr
