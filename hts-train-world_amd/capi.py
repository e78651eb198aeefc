"""The reference's per-utterance operator interface over host (numpy) arrays.

Calls the WORLD C ABI exported by libworld_mi355.so exactly as the reference's CLIs call
libworld.a (externs/WORLD_v2/test/analysis.cpp:93-203, test/synth.cpp:103-106): option
structs by pointer, caller-allocated outputs, ``double**`` rows for sp/ap.  Same function
names, argument order and meaning as the reference headers
(externs/WORLD_v2/src/world/{dio,harvest,stonemask,cheaptrick,d4c,synthesis}.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .world import LIB_PATH, load_library

_dp = C.POINTER(C.c_double)


class DioOption(C.Structure):            # dio.h:16-23
    _fields_ = [("f0_floor", C.c_double), ("f0_ceil", C.c_double), ("channels_in_octave", C.c_double),
                ("frame_period", C.c_double), ("speed", C.c_int), ("allowed_range", C.c_double)]


class HarvestOption(C.Structure):        # harvest.h:16-20
    _fields_ = [("f0_floor", C.c_double), ("f0_ceil", C.c_double), ("frame_period", C.c_double)]


class CheapTrickOption(C.Structure):     # cheaptrick.h:16-20
    _fields_ = [("q1", C.c_double), ("f0_floor", C.c_double), ("fft_size", C.c_int)]


class D4COption(C.Structure):            # d4c.h:16-18
    _fields_ = [("threshold", C.c_double)]


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def _rows(a):
    n = a.shape[0]
    arr = (_dp * n)()
    base, stride = a.ctypes.data, a.strides[0]
    for i in range(n):
        arr[i] = C.cast(base + i * stride, _dp)
    return arr


_L = None


def _lib():
    global _L
    if _L is None:
        load_library()
        L = C.CDLL(LIB_PATH)
        L.GetSamplesForDIO.restype = C.c_int
        L.GetSamplesForDIO.argtypes = [C.c_int, C.c_int, C.c_double]
        L.GetSamplesForHarvest.restype = C.c_int
        L.GetSamplesForHarvest.argtypes = [C.c_int, C.c_int, C.c_double]
        L.Dio.argtypes = [_dp, C.c_int, C.c_int, C.POINTER(DioOption), _dp, _dp]
        L.Harvest.argtypes = [_dp, C.c_int, C.c_int, C.POINTER(HarvestOption), _dp, _dp]
        L.StoneMask.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]
        L.CheapTrick.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.POINTER(CheapTrickOption),
                                 C.POINTER(_dp)]
        L.InitializeCheapTrickOption.argtypes = [C.c_int, C.POINTER(CheapTrickOption)]
        L.GetFFTSizeForCheapTrick.restype = C.c_int
        L.GetFFTSizeForCheapTrick.argtypes = [C.c_int, C.POINTER(CheapTrickOption)]
        L.GetF0FloorForCheapTrick.restype = C.c_double
        L.GetF0FloorForCheapTrick.argtypes = [C.c_int, C.c_int]
        L.D4C.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.POINTER(D4COption), C.POINTER(_dp)]
        L.Synthesis.argtypes = [_dp, C.c_int, C.POINTER(_dp), C.POINTER(_dp), C.c_int, C.c_double, C.c_int,
                                C.c_int, _dp]
        _L = L
    return _L


def dio(x, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0, channels_in_octave=2.0, speed=1,
        allowed_range=0.1):
    """Dio (dio.h:38-41) -> (temporal_positions, f0)."""
    L, x = _lib(), _c(x)
    opt = DioOption()
    L.InitializeDioOption(C.byref(opt))
    opt.frame_period, opt.f0_floor, opt.f0_ceil = frame_period, f0_floor, f0_ceil
    opt.channels_in_octave, opt.speed, opt.allowed_range = channels_in_octave, speed, allowed_range
    nf = L.GetSamplesForDIO(fs, len(x), frame_period)
    t, f0 = np.zeros(nf), np.zeros(nf)
    L.Dio(_p(x), len(x), fs, C.byref(opt), _p(t), _p(f0))
    return t, f0


def harvest(x, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0):
    """Harvest (harvest.h:35-37) -> (temporal_positions, f0)."""
    L, x = _lib(), _c(x)
    opt = HarvestOption()
    L.InitializeHarvestOption(C.byref(opt))
    opt.frame_period, opt.f0_floor, opt.f0_ceil = frame_period, f0_floor, f0_ceil
    nf = L.GetSamplesForHarvest(fs, len(x), frame_period)
    t, f0 = np.zeros(nf), np.zeros(nf)
    L.Harvest(_p(x), len(x), fs, C.byref(opt), _p(t), _p(f0))
    return t, f0


def stonemask(x, fs, temporal_positions, f0):
    """StoneMask (stonemask.h:27-29) -> refined_f0."""
    L, x, t, f0 = _lib(), _c(x), _c(temporal_positions), _c(f0)
    out = np.zeros(len(f0))
    L.StoneMask(_p(x), len(x), fs, _p(t), _p(f0), len(f0), _p(out))
    return out


def cheaptrick_fft_size(fs, f0_floor=71.0):
    opt = CheapTrickOption()
    opt.f0_floor = f0_floor
    return _lib().GetFFTSizeForCheapTrick(fs, C.byref(opt))


def cheaptrick(x, fs, temporal_positions, f0, q1=-0.15, fft_size=None):
    """CheapTrick (cheaptrick.h:38-40) -> spectrogram [frames, fft_size/2+1]."""
    L, x, t, f0 = _lib(), _c(x), _c(temporal_positions), _c(f0)
    opt = CheapTrickOption()
    L.InitializeCheapTrickOption(fs, C.byref(opt))
    opt.q1 = q1
    if fft_size:
        opt.fft_size = fft_size
    sp = np.zeros((len(f0), opt.fft_size // 2 + 1))
    L.CheapTrick(_p(x), len(x), fs, _p(t), _p(f0), len(f0), C.byref(opt), _rows(sp))
    return sp


def d4c(x, fs, temporal_positions, f0, fft_size, threshold=0.85):
    """D4C (d4c.h:35-37) -> aperiodicity [frames, fft_size/2+1]."""
    L, x, t, f0 = _lib(), _c(x), _c(temporal_positions), _c(f0)
    opt = D4COption()
    opt.threshold = threshold
    ap = np.zeros((len(f0), fft_size // 2 + 1))
    L.D4C(_p(x), len(x), fs, _p(t), _p(f0), len(f0), fft_size, C.byref(opt), _rows(ap))
    return ap


def synthesis(f0, spectrogram, aperiodicity, fft_size, frame_period, fs, y_length=None):
    """Synthesis (synthesis.h:30-32) -> y."""
    L, f0, sp, ap = _lib(), _c(f0), _c(spectrogram), _c(aperiodicity)
    if y_length is None:
        y_length = int((len(f0) - 1) * frame_period / 1000.0 * fs) + 1      # test/synth.cpp:259
    y = np.zeros(y_length)
    L.Synthesis(_p(f0), len(f0), _rows(sp), _rows(ap), fft_size, frame_period, fs, y_length, _p(y))
    return y


# ---- world/codec.h -------------------------------------------------------------------------------
def get_number_of_aperiodicities(fs):
    L = _lib()
    L.GetNumberOfAperiodicities.restype = C.c_int
    return int(L.GetNumberOfAperiodicities(int(fs)))


def code_spectral_envelope(spectrogram, fs, fft_size, number_of_dimensions):
    """CodeSpectralEnvelope (codec.h:69-72) -> [f0_length][number_of_dimensions]."""
    L, sp = _lib(), _c(spectrogram)
    out = np.zeros((sp.shape[0], number_of_dimensions))
    L.CodeSpectralEnvelope(_rows(sp), sp.shape[0], int(fs), int(fft_size), int(number_of_dimensions), _rows(out))
    return out


def decode_spectral_envelope(coded, fs, fft_size):
    """DecodeSpectralEnvelope (codec.h:86-88) -> [f0_length][fft_size/2+1]."""
    L, cd = _lib(), _c(coded)
    out = np.zeros((cd.shape[0], fft_size // 2 + 1))
    L.DecodeSpectralEnvelope(_rows(cd), cd.shape[0], int(fs), int(fft_size), cd.shape[1], _rows(out))
    return out


def code_aperiodicity(aperiodicity, fs, fft_size):
    """CodeAperiodicity (codec.h:38-39) -> [f0_length][GetNumberOfAperiodicities(fs)]."""
    L, ap = _lib(), _c(aperiodicity)
    nap = get_number_of_aperiodicities(fs)
    out = np.zeros((ap.shape[0], nap))
    L.CodeAperiodicity(_rows(ap), ap.shape[0], int(fs), int(fft_size), nap, _rows(out))
    return out


def decode_aperiodicity(coded, fs, fft_size):
    """DecodeAperiodicity; arguments in the order of the reference's DEFINITION (codec.cpp:237-238)."""
    L, cd = _lib(), _c(coded)
    out = np.zeros((cd.shape[0], fft_size // 2 + 1))
    L.DecodeAperiodicity(_rows(cd), cd.shape[0], int(fs), cd.shape[1], int(fft_size), _rows(out))
    return out
