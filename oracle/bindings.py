"""ctypes bindings for the parity oracle and (when built) the compiled reference.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.

``Oracle``    -> oracle/liboracle.so        (this repo's plain-C restatement)
``Reference`` -> oracle/_ref/libworld_ref.so (the real reference, compiled by
                 ``make -C oracle ref`` from /root/reference where it lies)

Both expose the same numpy-level methods so tests can swap one for the other.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def build(ref: bool = True) -> None:
    """Compile liboracle.so (always) and _ref (when the reference tree is present)."""
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir("/root/reference/externs/WORLD_v2/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class Oracle:
    kind = "port"

    def __init__(self, path: str | None = None):
        # ORACLE_LIB selects another build of the same sources (make -C oracle asan: tests/test_sanitizers.py)
        path = path or os.environ.get("ORACLE_LIB") or os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        L.orc_dio_samples.restype = C.c_int
        L.orc_dio_samples.argtypes = [C.c_int, C.c_int, C.c_double]
        L.orc_harvest_samples.restype = C.c_int
        L.orc_harvest_samples.argtypes = [C.c_int, C.c_int, C.c_double]
        L.orc_dio.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                              C.c_double, C.c_int, C.c_double, _dp, _dp]
        L.orc_harvest.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.orc_stonemask.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]
        L.orc_cheaptrick.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_int, _dp]
        L.orc_cheaptrick_fft_size.restype = C.c_int
        L.orc_cheaptrick_fft_size.argtypes = [C.c_int, C.c_double]
        L.orc_d4c.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp]
        L.orc_synthesis.argtypes = [_dp, C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, _dp]
        L.orc_randn_table.argtypes = [_dp, C.c_int]
        L.orc_randn_table_u32.argtypes = [C.POINTER(C.c_uint32), C.c_int]
        L.orc_interp1.argtypes = [_dp, _dp, C.c_int, _dp, C.c_int, _dp]
        L.orc_interp1q.argtypes = [C.c_double, C.c_double, _dp, C.c_int, _dp, C.c_int, _dp]
        L.orc_decimate.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.orc_nuttall.argtypes = [C.c_int, _dp]
        L.orc_dc_correction.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.orc_linear_smoothing.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.orc_fft_r2c.argtypes = [_dp, C.c_int, _dp, _dp]
        L.orc_fft_c2r.argtypes = [_dp, _dp, C.c_int, _dp]
        L.orc_min_phase.argtypes = [_dp, C.c_int, _dp, _dp]
        L.orc_matlab_round.restype = C.c_int
        L.orc_matlab_round.argtypes = [C.c_double]
        L.orc_suitable_fft_size.restype = C.c_int
        L.orc_suitable_fft_size.argtypes = [C.c_int]

    # -- public API mirror -------------------------------------------------
    def dio(self, x, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0,
            channels_in_octave=2.0, speed=1, allowed_range=0.1):
        x = _c(x)
        nf = self.lib.orc_dio_samples(fs, len(x), frame_period)
        t, f0 = np.zeros(nf), np.zeros(nf)
        self.lib.orc_dio(_p(x), len(x), fs, f0_floor, f0_ceil, channels_in_octave,
                         frame_period, speed, allowed_range, _p(t), _p(f0))
        return t, f0

    def harvest(self, x, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0):
        x = _c(x)
        nf = self.lib.orc_harvest_samples(fs, len(x), frame_period)
        t, f0 = np.zeros(nf), np.zeros(nf)
        self.lib.orc_harvest(_p(x), len(x), fs, f0_floor, f0_ceil, frame_period, _p(t), _p(f0))
        return t, f0

    def stonemask(self, x, fs, t, f0):
        x, t, f0 = _c(x), _c(t), _c(f0)
        out = np.zeros(len(f0))
        self.lib.orc_stonemask(_p(x), len(x), fs, _p(t), _p(f0), len(f0), _p(out))
        return out

    # -- feature codec (codec.cpp) -------------------------------------------
    def num_aperiodicities(self, fs):
        return int(self.lib.orc_num_aperiodicities(int(fs)))

    def code_aperiodicity(self, ap, fs, fft_size):
        ap = _c(ap)
        nap = self.num_aperiodicities(fs)
        out = np.zeros((ap.shape[0], nap))
        self.lib.orc_code_aperiodicity(_p(ap), ap.shape[0], int(fs), int(fft_size), nap, _p(out))
        return out

    def decode_aperiodicity(self, coded, fs, fft_size):
        coded = _c(coded)
        out = np.zeros((coded.shape[0], fft_size // 2 + 1))
        self.lib.orc_decode_aperiodicity(_p(coded), coded.shape[0], int(fs), coded.shape[1], int(fft_size), _p(out))
        return out

    def code_spectral_envelope(self, sp, fs, fft_size, ndim):
        sp = _c(sp)
        out = np.zeros((sp.shape[0], ndim))
        self.lib.orc_code_spectral_envelope(_p(sp), sp.shape[0], int(fs), int(fft_size), int(ndim), _p(out))
        return out

    def decode_spectral_envelope(self, coded, fs, fft_size):
        coded = _c(coded)
        out = np.zeros((coded.shape[0], fft_size // 2 + 1))
        self.lib.orc_decode_spectral_envelope(_p(coded), coded.shape[0], int(fs), int(fft_size), coded.shape[1],
                                              _p(out))
        return out

    def freqt(self, c1, m2, a):
        c1 = _c(c1)
        out = np.zeros(m2 + 1)
        self.lib.orc_freqt.argtypes = [_dp, C.c_int, _dp, C.c_int, C.c_double]
        self.lib.orc_freqt(_p(c1), len(c1) - 1, _p(out), int(m2), float(a))
        return out

    def mgc2sp(self, mgc, alpha, fft_size, nbins=None):
        """log-amplitude spectrum (the x output of the CLI's mgc2sp with gamma 0); len(mgc) = order + 1"""
        mgc = _c(mgc)
        nbins = fft_size if nbins is None else nbins
        out = np.zeros(nbins)
        self.lib.orc_mgc2sp.argtypes = [_dp, C.c_int, C.c_double, C.c_int, C.c_int, _dp]
        self.lib.orc_mgc2sp(_p(mgc), len(mgc) - 1, float(alpha), int(fft_size), int(nbins), _p(out))
        return out

    def recipe_decode(self, lf0, mgc, bap, fs, fft_size):
        """test/synth.cpp:151-256: float32 lf0 / mgc / bap -> f0, sp, ap (ap bins >= order defined as 0)"""
        lf0 = np.ascontiguousarray(lf0, dtype=np.float32)
        mgc = np.ascontiguousarray(mgc, dtype=np.float32)
        bap = np.ascontiguousarray(bap, dtype=np.float32)
        nf, w = len(lf0), fft_size // 2 + 1
        f0, sp, ap = np.zeros(nf), np.zeros((nf, w)), np.zeros((nf, w))
        fp_ = C.POINTER(C.c_float)
        self.lib.orc_recipe_decode.argtypes = [fp_, fp_, fp_, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]
        self.lib.orc_recipe_decode(lf0.ctypes.data_as(fp_), mgc.ctypes.data_as(fp_), bap.ctypes.data_as(fp_), nf,
                                   int(fs), int(fft_size), mgc.shape[1], bap.shape[1], _p(f0), _p(sp), _p(ap))
        return f0, sp, ap

    def cheaptrick_fft_size(self, fs, f0_floor=71.0):
        return self.lib.orc_cheaptrick_fft_size(fs, f0_floor)

    def cheaptrick(self, x, fs, t, f0, q1=-0.15, fft_size=None):
        x, t, f0 = _c(x), _c(t), _c(f0)
        fft_size = fft_size or self.cheaptrick_fft_size(fs)
        sp = np.zeros((len(f0), fft_size // 2 + 1))
        self.lib.orc_cheaptrick(_p(x), len(x), fs, _p(t), _p(f0), len(f0), q1, fft_size, _p(sp))
        return sp

    def d4c(self, x, fs, t, f0, fft_size, threshold=0.85):
        x, t, f0 = _c(x), _c(t), _c(f0)
        ap = np.zeros((len(f0), fft_size // 2 + 1))
        self.lib.orc_d4c(_p(x), len(x), fs, _p(t), _p(f0), len(f0), fft_size, threshold, _p(ap))
        return ap

    def synthesis(self, f0, sp, ap, fft_size, frame_period, fs, y_length=None):
        f0, sp, ap = _c(f0), _c(sp), _c(ap)
        if y_length is None:
            y_length = int((len(f0) - 1) * frame_period / 1000.0 * fs) + 1   # synth.cpp:259
        y = np.zeros(y_length)
        self.lib.orc_synthesis(_p(f0), len(f0), _p(sp), _p(ap), fft_size, frame_period, fs,
                               y_length, _p(y))
        return y

    # -- cmp composition (window.pl, addhtkheader.pl) --------------------------
    def window_stream(self, data, windows):
        """data: float32 [T][dim]; windows: list of coefficient lists (the window files without the size)."""
        d = np.ascontiguousarray(data, dtype=np.float32)
        T, dim = d.shape
        ws = [np.array([len(w)] + list(w), dtype=np.float64) for w in windows]
        arr = (_dp * len(ws))(*[w.ctypes.data_as(_dp) for w in ws])
        out = np.zeros((T, len(ws) * dim), dtype=np.float32)
        self.lib.orc_window_stream(d.ctypes.data_as(C.c_void_p), T, dim, len(ws), arr, out.ctypes.data_as(C.c_void_p))
        return out

    def htk_header(self, nframes, samprate, frameshift, bytes_per_frame, kind=9):
        buf = (C.c_ubyte * 12)()
        self.lib.orc_htk_header(int(nframes), int(samprate), int(frameshift), int(bytes_per_frame), int(kind), buf)
        return bytes(buf)

    # -- vibrato (Extract.py; world_oracle_vibrato.c, parity unpinned) ---------
    def lowess(self, y, frac=2.0 / 3.0, it=20):
        y = _c(y)
        out = np.zeros(len(y))
        self.lib.orc_lowess.argtypes = [_dp, C.c_int, C.c_double, C.c_int, _dp]
        self.lib.orc_lowess(_p(y), len(y), frac, it, _p(out))
        return out

    def get_vibrate(self, f):
        f = _c(f)
        t = np.zeros((2 * len(f) + 2, 2))
        self.lib.orc_get_vibrate.restype = C.c_int
        self.lib.orc_get_vibrate.argtypes = [_dp, C.c_int, _dp]
        n = self.lib.orc_get_vibrate(_p(f), len(f), _p(t))
        return t[:n]

    def vibrato(self, lf0, segments):
        """lf0: float32 [T] as the lf0 file holds it; segments: [(start_frame, end_frame, pitch_hz)].
        Returns (vib, lf0_2col) float32 [T][2] after soprLog, and the number of runs."""
        e = np.exp(np.asarray(lf0, dtype=np.float32).astype(np.float64))
        f0 = np.where(e < 1.0, 0.0, e)                                  # soprExp, Extract.py:98-108
        T = len(f0)
        ip = C.POINTER(C.c_int)
        ss = np.ascontiguousarray([s[0] for s in segments], dtype=np.int32)
        se = np.ascontiguousarray([s[1] for s in segments], dtype=np.int32)
        sp = np.ascontiguousarray([s[2] for s in segments], dtype=np.float64)
        vib, df0 = np.zeros((T, 2)), np.zeros((T, 2))
        self.lib.orc_vibrato.restype = C.c_int
        self.lib.orc_vibrato.argtypes = [_dp, C.c_int, ip, ip, _dp, C.c_int, _dp, _dp]
        runs = self.lib.orc_vibrato(_p(f0), T, ss.ctypes.data_as(ip), se.ctypes.data_as(ip), _p(sp), len(ss), _p(vib), _p(df0))
        fp_ = C.POINTER(C.c_float)
        self.lib.orc_sopr_log.argtypes = [_dp, C.c_int, fp_]
        o1, o2 = np.zeros((T, 2), dtype=np.float32), np.zeros((T, 2), dtype=np.float32)
        self.lib.orc_sopr_log(_p(vib), 2 * T, o1.ctypes.data_as(fp_))
        self.lib.orc_sopr_log(_p(df0), 2 * T, o2.ctypes.data_as(fp_))
        return o1, o2, runs

    # -- primitives ----------------------------------------------------------
    def randn_table(self, n):
        out = np.zeros(n)
        self.lib.orc_randn_table(_p(out), n)
        return out

    def randn_table_u32(self, n):
        out = np.zeros(n, dtype=np.uint32)
        self.lib.orc_randn_table_u32(out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out

    def interp1(self, x, y, xi):
        x, y, xi = _c(x), _c(y), _c(xi)
        yi = np.zeros(len(xi))
        self.lib.orc_interp1(_p(x), _p(y), len(x), _p(xi), len(xi), _p(yi))
        return yi

    def interp1q(self, x0, dx, y, xi):
        y, xi = _c(y), _c(xi)
        yi = np.zeros(len(xi))
        self.lib.orc_interp1q(x0, dx, _p(y), len(y), _p(xi), len(xi), _p(yi))
        return yi

    def decimate(self, x, r):
        x = _c(x)
        y = np.zeros((len(x) - 1) // r + 1 + 32)
        self.lib.orc_decimate(_p(x), len(x), r, _p(y))
        return y[: (len(x) - 1) // r + 1]

    def nuttall(self, n):
        w = np.zeros(n)
        self.lib.orc_nuttall(n, _p(w))
        return w

    def dc_correction(self, spec, f0, fs, fft_size):
        spec = _c(spec)
        out = spec.copy()
        self.lib.orc_dc_correction(_p(spec), f0, fs, fft_size, _p(out))
        return out

    def linear_smoothing(self, spec, width, fs, fft_size):
        spec = _c(spec)
        out = np.zeros(fft_size // 2 + 1)
        self.lib.orc_linear_smoothing(_p(spec), width, fs, fft_size, _p(out))
        return out

    def fft_r2c(self, x):
        x = _c(x)
        n = len(x)
        re, im = np.zeros(n // 2 + 1), np.zeros(n // 2 + 1)
        self.lib.orc_fft_r2c(_p(x), n, _p(re), _p(im))
        return re + 1j * im

    def fft_c2r(self, spec, n):
        re, im = _c(spec.real), _c(spec.imag)
        x = np.zeros(n)
        self.lib.orc_fft_c2r(_p(re), _p(im), n, _p(x))
        return x

    def min_phase(self, log_spec, fft_size):
        ls = _c(log_spec)
        re, im = np.zeros(fft_size // 2 + 1), np.zeros(fft_size // 2 + 1)
        self.lib.orc_min_phase(_p(ls), fft_size, _p(re), _p(im))
        return re + 1j * im


# ---- the real reference ------------------------------------------------------
class _DioOption(C.Structure):          # world/dio.h:16-23
    _fields_ = [("f0_floor", C.c_double), ("f0_ceil", C.c_double),
                ("channels_in_octave", C.c_double), ("frame_period", C.c_double),
                ("speed", C.c_int), ("allowed_range", C.c_double)]


class _HarvestOption(C.Structure):      # world/harvest.h:16-20
    _fields_ = [("f0_floor", C.c_double), ("f0_ceil", C.c_double), ("frame_period", C.c_double)]


class _CheapTrickOption(C.Structure):   # world/cheaptrick.h:16-20
    _fields_ = [("q1", C.c_double), ("f0_floor", C.c_double), ("fft_size", C.c_int)]


class _D4COption(C.Structure):          # world/d4c.h:16-18
    _fields_ = [("threshold", C.c_double)]


def _rows(a: np.ndarray):
    """double** over the rows of a C-contiguous 2-D array."""
    n = a.shape[0]
    arr = (_dp * n)()
    base, stride = a.ctypes.data, a.strides[0]
    for i in range(n):
        arr[i] = C.cast(base + i * stride, _dp)
    return arr


class WorldCApi:
    """Caller of WORLD's public C ABI; works on any .so exporting it (the
    compiled reference, or this repo's libworld_mi355.so drop-in)."""
    kind = "reference"

    def __init__(self, path: str):
        self.lib = L = C.CDLL(path)
        L.GetSamplesForDIO.restype = C.c_int
        L.GetSamplesForDIO.argtypes = [C.c_int, C.c_int, C.c_double]
        L.Dio.argtypes = [_dp, C.c_int, C.c_int, C.POINTER(_DioOption), _dp, _dp]
        L.InitializeDioOption.argtypes = [C.POINTER(_DioOption)]
        L.StoneMask.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]
        L.CheapTrick.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int,
                                 C.POINTER(_CheapTrickOption), C.POINTER(_dp)]
        L.InitializeCheapTrickOption.argtypes = [C.c_int, C.POINTER(_CheapTrickOption)]
        L.GetFFTSizeForCheapTrick.restype = C.c_int
        L.GetFFTSizeForCheapTrick.argtypes = [C.c_int, C.POINTER(_CheapTrickOption)]
        L.GetF0FloorForCheapTrick.restype = C.c_double
        L.GetF0FloorForCheapTrick.argtypes = [C.c_int, C.c_int]
        L.D4C.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int,
                          C.POINTER(_D4COption), C.POINTER(_dp)]
        L.InitializeD4COption.argtypes = [C.POINTER(_D4COption)]
        L.Synthesis.argtypes = [_dp, C.c_int, C.POINTER(_dp), C.POINTER(_dp), C.c_int,
                                C.c_double, C.c_int, C.c_int, _dp]
        self.has_harvest = hasattr(L, "Harvest")
        if self.has_harvest:
            L.GetSamplesForHarvest.restype = C.c_int
            L.GetSamplesForHarvest.argtypes = [C.c_int, C.c_int, C.c_double]
            L.Harvest.argtypes = [_dp, C.c_int, C.c_int, C.POINTER(_HarvestOption), _dp, _dp]
            L.InitializeHarvestOption.argtypes = [C.POINTER(_HarvestOption)]

    def dio(self, x, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0,
            channels_in_octave=2.0, speed=1, allowed_range=0.1):
        x = _c(x)
        opt = _DioOption()
        self.lib.InitializeDioOption(C.byref(opt))
        opt.frame_period, opt.f0_floor, opt.f0_ceil = frame_period, f0_floor, f0_ceil
        opt.channels_in_octave, opt.speed, opt.allowed_range = channels_in_octave, speed, allowed_range
        nf = self.lib.GetSamplesForDIO(fs, len(x), frame_period)
        t, f0 = np.zeros(nf), np.zeros(nf)
        self.lib.Dio(_p(x), len(x), fs, C.byref(opt), _p(t), _p(f0))
        return t, f0

    def harvest(self, x, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0):
        x = _c(x)
        opt = _HarvestOption()
        self.lib.InitializeHarvestOption(C.byref(opt))
        opt.frame_period, opt.f0_floor, opt.f0_ceil = frame_period, f0_floor, f0_ceil
        nf = self.lib.GetSamplesForHarvest(fs, len(x), frame_period)
        t, f0 = np.zeros(nf), np.zeros(nf)
        self.lib.Harvest(_p(x), len(x), fs, C.byref(opt), _p(t), _p(f0))
        return t, f0

    def stonemask(self, x, fs, t, f0):
        x, t, f0 = _c(x), _c(t), _c(f0)
        out = np.zeros(len(f0))
        self.lib.StoneMask(_p(x), len(x), fs, _p(t), _p(f0), len(f0), _p(out))
        return out

    def cheaptrick_fft_size(self, fs, f0_floor=71.0):
        opt = _CheapTrickOption()
        opt.f0_floor = f0_floor
        return self.lib.GetFFTSizeForCheapTrick(fs, C.byref(opt))

    def cheaptrick(self, x, fs, t, f0, q1=-0.15, fft_size=None):
        x, t, f0 = _c(x), _c(t), _c(f0)
        opt = _CheapTrickOption()
        self.lib.InitializeCheapTrickOption(fs, C.byref(opt))
        opt.q1 = q1
        if fft_size:
            opt.fft_size = fft_size
        sp = np.zeros((len(f0), opt.fft_size // 2 + 1))
        self.lib.CheapTrick(_p(x), len(x), fs, _p(t), _p(f0), len(f0), C.byref(opt), _rows(sp))
        return sp

    def d4c(self, x, fs, t, f0, fft_size, threshold=0.85):
        x, t, f0 = _c(x), _c(t), _c(f0)
        opt = _D4COption()
        opt.threshold = threshold
        ap = np.zeros((len(f0), fft_size // 2 + 1))
        self.lib.D4C(_p(x), len(x), fs, _p(t), _p(f0), len(f0), fft_size, C.byref(opt), _rows(ap))
        return ap

    def synthesis(self, f0, sp, ap, fft_size, frame_period, fs, y_length=None):
        f0, sp, ap = _c(f0), _c(sp), _c(ap)
        if y_length is None:
            y_length = int((len(f0) - 1) * frame_period / 1000.0 * fs) + 1
        y = np.zeros(y_length)
        self.lib.Synthesis(_p(f0), len(f0), _rows(sp), _rows(ap), fft_size, frame_period, fs,
                           y_length, _p(y))
        return y

    # -- feature codec (world/codec.h) ---------------------------------------
    def num_aperiodicities(self, fs):
        self.lib.GetNumberOfAperiodicities.restype = C.c_int
        return int(self.lib.GetNumberOfAperiodicities(int(fs)))

    def code_aperiodicity(self, ap, fs, fft_size):
        ap = _c(ap)
        nap = self.num_aperiodicities(fs)
        out = np.zeros((ap.shape[0], nap))
        self.lib.CodeAperiodicity(_rows(ap), ap.shape[0], int(fs), int(fft_size), nap, _rows(out))
        return out

    def decode_aperiodicity(self, coded, fs, fft_size):
        """Arguments follow the DEFINITION's order (codec.cpp:237-238): (.., fs, nap, fft_size, ..)."""
        coded = _c(coded)
        out = np.zeros((coded.shape[0], fft_size // 2 + 1))
        self.lib.DecodeAperiodicity(_rows(coded), coded.shape[0], int(fs), coded.shape[1], int(fft_size), _rows(out))
        return out

    def code_spectral_envelope(self, sp, fs, fft_size, ndim):
        sp = _c(sp)
        out = np.zeros((sp.shape[0], ndim))
        self.lib.CodeSpectralEnvelope(_rows(sp), sp.shape[0], int(fs), int(fft_size), int(ndim), _rows(out))
        return out

    def decode_spectral_envelope(self, coded, fs, fft_size):
        coded = _c(coded)
        out = np.zeros((coded.shape[0], fft_size // 2 + 1))
        self.lib.DecodeSpectralEnvelope(_rows(coded), coded.shape[0], int(fs), int(fft_size), coded.shape[1],
                                        _rows(out))
        return out


class Reference(WorldCApi):
    """The compiled reference plus the primitives it happens to export."""

    PATH = os.path.join(HERE, "_ref", "libworld_ref.so")
    PATH_O2 = os.path.join(HERE, "_ref", "libworld_ref_O2.so")      # same sources at -O2: timing only (bench.py)

    @classmethod
    def available(cls, o2: bool = False) -> bool:
        return os.path.exists(cls.PATH_O2 if o2 else cls.PATH)

    def __init__(self, o2: bool = False):
        super().__init__(self.PATH_O2 if o2 else self.PATH)
        L = self.lib
        L.randn.restype = C.c_double
        L.interp1.argtypes = [_dp, _dp, C.c_int, _dp, C.c_int, _dp]
        L.interp1Q.argtypes = [C.c_double, C.c_double, _dp, C.c_int, _dp, C.c_int, _dp]
        L.decimate.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.NuttallWindow.argtypes = [C.c_int, _dp]
        L.DCCorrection.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.LinearSmoothing.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.matlab_round.restype = C.c_int
        L.matlab_round.argtypes = [C.c_double]
        L.GetSuitableFFTSize.restype = C.c_int
        L.GetSuitableFFTSize.argtypes = [C.c_int]

    def randn_table(self, n):
        self.lib.randn_reseed()
        return np.array([self.lib.randn() for _ in range(n)])

    def interp1(self, x, y, xi):
        x, y, xi = _c(x), _c(y), _c(xi)
        yi = np.zeros(len(xi))
        self.lib.interp1(_p(x), _p(y), len(x), _p(xi), len(xi), _p(yi))
        return yi

    def interp1q(self, x0, dx, y, xi):
        y, xi = _c(y), _c(xi)
        yi = np.zeros(len(xi))
        self.lib.interp1Q(x0, dx, _p(y), len(y), _p(xi), len(xi), _p(yi))
        return yi

    def decimate(self, x, r):
        x = _c(x)
        y = np.zeros((len(x) - 1) // r + 1 + 32)
        self.lib.decimate(_p(x), len(x), r, _p(y))
        return y[: (len(x) - 1) // r + 1]

    def nuttall(self, n):
        w = np.zeros(n)
        self.lib.NuttallWindow(n, _p(w))
        return w

    def dc_correction(self, spec, f0, fs, fft_size):
        spec = _c(spec)
        out = spec.copy()
        self.lib.DCCorrection(_p(spec), f0, fs, fft_size, _p(out))
        return out

    def linear_smoothing(self, spec, width, fs, fft_size):
        spec = _c(spec)
        out = np.zeros(fft_size // 2 + 1)
        self.lib.LinearSmoothing(_p(spec), width, fs, fft_size, _p(out))
        return out


class SptkReference:
    """The CLIs' SPTK port (test/sptkfunctions.cpp) compiled as it is: mgc2sp / freqt, C++-mangled names."""

    PATH = os.path.join(HERE, "_ref", "libsptk_ref.so")

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = L = C.CDLL(self.PATH)
        self._mgc2sp = L._Z6mgc2spPdiddS_S_i           # void mgc2sp(double*, int, double, double, double*, double*, int)
        self._mgc2sp.argtypes = [_dp, C.c_int, C.c_double, C.c_double, _dp, _dp, C.c_int]
        self._freqt = L._Z5freqtPdiS_id                # void freqt(double*, int, double*, int, double)
        self._freqt.argtypes = [_dp, C.c_int, _dp, C.c_int, C.c_double]

    def freqt(self, c1, m2, a):
        c1 = _c(c1).copy()
        out = np.zeros(m2 + 1)
        self._freqt(_p(c1), len(c1) - 1, _p(out), int(m2), float(a))
        return out

    def mgc2sp(self, mgc, alpha, fft_size):
        mgc = _c(mgc).copy()
        x, y = np.zeros(fft_size), np.zeros(fft_size)
        self._mgc2sp(_p(mgc), len(mgc) - 1, float(alpha), 0.0, _p(x), _p(y), int(fft_size))
        return x
