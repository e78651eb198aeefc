"""ctypes host layer over libworld_mi355.so (the C ABI declared in include/).

Mirrors the reference's per-utterance operator interface
(externs/WORLD_v2/src/world/{dio,stonemask,cheaptrick,d4c,synthesis,harvest}.h)
for numpy callers and adds ``WorldBatch`` for device-resident batches (torch
tensors on ``cuda``).  torch is used only for device memory and streams.

The library is mandatory: nothing here computes WORLD on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# WORLD_MI355_LIB names another build of the same library (tools/ab_lib.sh: A/B runs that leave the in-tree file alone)
LIB_PATH = os.environ.get("WORLD_MI355_LIB") or os.path.join(HERE, "libworld_mi355.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

ERRORS = {1: "HIP error", 2: "bad argument", 3: "unsupported fft_size", 4: "no HIP device",
          5: "unsupported configuration"}


class WorldParams(C.Structure):
    """include/world_mi355.h: WorldMi355Params."""
    _fields_ = [("fs", C.c_int), ("frame_period", C.c_double), ("f0_floor", C.c_double),
                ("f0_ceil", C.c_double), ("channels_in_octave", C.c_double), ("speed", C.c_int),
                ("allowed_range", C.c_double), ("q1", C.c_double), ("fft_size", C.c_int),
                ("d4c_threshold", C.c_double)]


def build_library() -> None:
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "csrc"), "-j8"])


_lib = None


def load_library():
    """Load libworld_mi355.so; raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950); "
                           "this package has no CPU path")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.WorldMi355LastError.restype = C.c_char_p
    L.WorldMi355DefaultParams.argtypes = [C.c_int, C.c_double, C.POINTER(WorldParams)]
    L.WorldMi355CreateContext.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.WorldMi355DestroyContext.argtypes = [vp]
    L.WorldMi355SetStream.argtypes = [vp, vp]
    L.WorldMi355Synchronize.argtypes = [vp]
    L.WorldMi355CreateBatch.argtypes = [vp, C.POINTER(WorldParams), C.c_int, _ip, _ip, _ip, C.POINTER(vp)]
    L.WorldMi355DestroyBatch.argtypes = [vp]
    for name in ("TotalSamples", "TotalFrames", "TotalOutputSamples"):
        f = getattr(L, "WorldMi355Batch" + name)
        f.restype = C.c_int64
        f.argtypes = [vp]
    L.WorldMi355BatchFftSize.argtypes = [vp]
    for name in ("SampleOffsets", "FrameOffsets", "OutputOffsets"):
        f = getattr(L, "WorldMi355Batch" + name)
        f.restype = C.POINTER(C.c_int64)
        f.argtypes = [vp]
    L.WorldMi355Dio.argtypes = [vp, vp, vp, vp]
    L.WorldMi355Harvest.argtypes = [vp, vp, vp, vp]
    L.WorldMi355StoneMask.argtypes = [vp, vp, vp, vp, vp]
    L.WorldMi355CheapTrick.argtypes = [vp, vp, vp, vp, vp]
    L.WorldMi355D4C.argtypes = [vp, vp, vp, vp, vp]
    L.WorldMi355Synthesis.argtypes = [vp, vp, vp, vp, vp]
    L.WorldMi355SamplesFromPcm16.argtypes = [vp, vp, vp]
    L.WorldMi355SamplesToPcm16.argtypes = [vp, vp, vp]
    L.WorldMi355Analyze.argtypes = [vp, vp, vp, vp, vp, vp]
    L.WorldMi355AnalyzeSynthesize.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.WorldMi355UtteranceStatus.argtypes = [vp, vp, vp, vp, vp, vp]
    L.WorldMi355Vibrato.argtypes = [vp, vp, _ip, _ip, _ip, _dp, vp, vp, _ip]
    L.WorldMi355GetNumberOfAperiodicities.argtypes = [C.c_int]
    L.WorldMi355CodeSpectralEnvelope.argtypes = [vp, vp, C.c_int, vp]
    L.WorldMi355DecodeSpectralEnvelope.argtypes = [vp, vp, C.c_int, vp]
    L.WorldMi355CodeAperiodicity.argtypes = [vp, vp, vp]
    L.WorldMi355DecodeAperiodicity.argtypes = [vp, vp, vp]
    L.WorldMi355RecipeFeatures.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.WorldMi355RecipeDecode.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.WorldMi355ComposeCmp.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.WorldMi355WriteFiles.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(C.c_size_t), C.c_int]
    L.WorldMi355HtkHeader.restype = None
    L.WorldMi355HtkHeader.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.WorldMi355TimingEnable.argtypes = [vp, C.c_int]
    L.WorldMi355TimingQuery.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    _lib = L
    return L


def _check(rc: int, where: str) -> None:
    if rc != 0:
        msg = load_library().WorldMi355LastError()
        raise RuntimeError(f"{where}: {ERRORS.get(rc, rc)} ({msg.decode() if msg else ''})")


def default_params(fs: int, frame_period: float = 5.0, **over) -> WorldParams:
    p = WorldParams()
    load_library().WorldMi355DefaultParams(fs, frame_period, C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


class Context:
    """One HIP stream + the universal randn table on one device."""

    def __init__(self, device: int | None = None, stream_ptr: int | None = None):
        L = load_library()
        h = C.c_void_p()
        _check(L.WorldMi355CreateContext(-1 if device is None else device,
                                         C.c_void_p(stream_ptr) if stream_ptr else None, C.byref(h)),
               "CreateContext")
        self.handle = h

    def synchronize(self):
        _check(load_library().WorldMi355Synchronize(self.handle), "Synchronize")

    def timing_enable(self, on: bool = True):
        """Record HIP events on the context's stream around every kernel launch."""
        _check(load_library().WorldMi355TimingEnable(self.handle, 1 if on else 0), "TimingEnable")

    def timing_query(self, kernel: str):
        """(total milliseconds, launches) of `kernel` since timing_enable(); synchronises."""
        ms, n = C.c_double(0.0), C.c_int(0)
        _check(load_library().WorldMi355TimingQuery(self.handle, kernel.encode(), C.byref(ms), C.byref(n)),
               "TimingQuery")
        return ms.value, n.value

    def close(self):
        if self.handle:
            load_library().WorldMi355DestroyContext(self.handle)
            self.handle = None


def _ints(a):
    if a is None:
        return None, None
    arr = np.ascontiguousarray(a, dtype=np.int32)
    return arr, arr.ctypes.data_as(_ip)


class WorldBatch:
    """A batch of utterances resident in HBM (torch cuda tensors, float64).

    x is the concatenation of the waveforms (see ``sample_offsets``); t, f0 have
    ``total_frames`` entries, sp/ap ``total_frames x (fft_size/2+1)``.
    """

    def __init__(self, ctx: Context, params: WorldParams, x_lengths=None, f0_lengths=None, y_lengths=None):
        L = load_library()
        self.ctx = ctx
        self.params = params
        n = len(x_lengths) if x_lengths is not None else len(f0_lengths)
        xa, xp = _ints(x_lengths)
        fa, fp = _ints(f0_lengths)
        ya, yp = _ints(y_lengths)
        h = C.c_void_p()
        _check(L.WorldMi355CreateBatch(ctx.handle, C.byref(params), n, xp, fp, yp, C.byref(h)), "CreateBatch")
        self.handle = h
        self.n_utt = n
        self.total_samples = L.WorldMi355BatchTotalSamples(h)
        self.total_frames = L.WorldMi355BatchTotalFrames(h)
        self.total_out = L.WorldMi355BatchTotalOutputSamples(h)
        self.fft_size = L.WorldMi355BatchFftSize(h)
        self.bins = self.fft_size // 2 + 1
        self.sample_offsets = np.ctypeslib.as_array(L.WorldMi355BatchSampleOffsets(h), (n + 1,)).copy()
        self.frame_offsets = np.ctypeslib.as_array(L.WorldMi355BatchFrameOffsets(h), (n + 1,)).copy()
        self.out_offsets = np.ctypeslib.as_array(L.WorldMi355BatchOutputOffsets(h), (n + 1,)).copy()

    @staticmethod
    def _p(t):
        assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch.float64", "cuda float64 contiguous"
        return C.c_void_p(t.data_ptr())

    def _new(self, *shape):
        import torch
        return torch.empty(*shape, dtype=torch.float64, device="cuda")

    def dio(self, x):
        t, f0 = self._new(self.total_frames), self._new(self.total_frames)
        _check(load_library().WorldMi355Dio(self.handle, self._p(x), self._p(t), self._p(f0)), "Dio")
        return t, f0

    def harvest(self, x):
        t, f0 = self._new(self.total_frames), self._new(self.total_frames)
        _check(load_library().WorldMi355Harvest(self.handle, self._p(x), self._p(t), self._p(f0)), "Harvest")
        return t, f0

    def stonemask(self, x, t, f0):
        out = self._new(self.total_frames)
        _check(load_library().WorldMi355StoneMask(self.handle, self._p(x), self._p(t), self._p(f0), self._p(out)),
               "StoneMask")
        return out

    def cheaptrick(self, x, t, f0, out=None):
        sp = out if out is not None else self._new(self.total_frames, self.bins)
        _check(load_library().WorldMi355CheapTrick(self.handle, self._p(x), self._p(t), self._p(f0), self._p(sp)),
               "CheapTrick")
        return sp

    def d4c(self, x, t, f0, out=None):
        ap = out if out is not None else self._new(self.total_frames, self.bins)
        _check(load_library().WorldMi355D4C(self.handle, self._p(x), self._p(t), self._p(f0), self._p(ap)), "D4C")
        return ap

    def samples_from_pcm16(self, pcm, out=None):
        """int16 cuda tensor [total_samples] (the wav payload) -> float64 samples s / 32768 (wavread)."""
        assert pcm.is_cuda and pcm.is_contiguous() and str(pcm.dtype) == "torch.int16" and pcm.numel() == self.total_samples
        x = out if out is not None else self._new(self.total_samples)
        _check(load_library().WorldMi355SamplesFromPcm16(self.handle, C.c_void_p(pcm.data_ptr()), self._p(x)),
               "SamplesFromPcm16")
        return x

    def samples_to_pcm16(self, y, out=None):
        """float64 y [total_out] -> int16 as wavwrite stores it: clamp(int(y * 32767)), truncating towards zero."""
        import torch
        pcm = out if out is not None else torch.empty(self.total_out, dtype=torch.int16, device="cuda")
        assert pcm.is_cuda and pcm.is_contiguous() and pcm.numel() == self.total_out
        _check(load_library().WorldMi355SamplesToPcm16(self.handle, self._p(y), C.c_void_p(pcm.data_ptr())),
               "SamplesToPcm16")
        return pcm

    def analyze(self, x, out=None):
        """Dio -> StoneMask -> CheapTrick -> D4C (test/analysis.cpp:243-390)."""
        if out is None:
            out = (self._new(self.total_frames), self._new(self.total_frames),
                   self._new(self.total_frames, self.bins), self._new(self.total_frames, self.bins))
        t, f0, sp, ap = out
        _check(load_library().WorldMi355Analyze(self.handle, self._p(x), self._p(t), self._p(f0), self._p(sp),
                                                self._p(ap)), "Analyze")
        return t, f0, sp, ap

    def analyze_synthesize(self, x, out=None, y=None):
        """analyze() then synthesize() of its own features as one call: the f0-only part of Synthesis overlaps
        CheapTrick and D4C on a second stream.  Returns (t, f0, sp, ap, y), bit-identical to the two calls."""
        if out is None:
            out = (self._new(self.total_frames), self._new(self.total_frames),
                   self._new(self.total_frames, self.bins), self._new(self.total_frames, self.bins))
        t, f0, sp, ap = out
        y = y if y is not None else self._new(self.total_out)
        _check(load_library().WorldMi355AnalyzeSynthesize(self.handle, self._p(x), self._p(t), self._p(f0),
                                                          self._p(sp), self._p(ap), self._p(y)), "AnalyzeSynthesize")
        return t, f0, sp, ap, y

    def synthesize(self, f0, sp, ap, out=None):
        y = out if out is not None else self._new(self.total_out)
        _check(load_library().WorldMi355Synthesis(self.handle, self._p(f0), self._p(sp), self._p(ap), self._p(y)),
               "Synthesis")
        return y

    def vibrato(self, lf0, segments):
        """The recipe's vibrato feature (data/scripts/Extract.py).  lf0: float32 cuda [total_frames];
        segments: per utterance a list of (start_frame, end_frame, note_pitch_hz).  Returns (vib, lf0_2col, n_too_long):
        float32 cuda tensors [total_frames][2] after the script's soprLog."""
        import torch
        assert lf0.dtype == torch.float32 and lf0.is_cuda and lf0.is_contiguous() and len(segments) == self.n_utt
        off = np.zeros(self.n_utt + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(s) for s in segments])
        flat = [seg for s in segments for seg in s]
        ss = np.ascontiguousarray([int(v[0]) for v in flat], dtype=np.int32)
        se = np.ascontiguousarray([int(v[1]) for v in flat], dtype=np.int32)
        sp_ = np.ascontiguousarray([float(v[2]) for v in flat], dtype=np.float64)
        vib = torch.empty(self.total_frames, 2, dtype=torch.float32, device="cuda")
        out = torch.empty(self.total_frames, 2, dtype=torch.float32, device="cuda")
        n = C.c_int(0)
        _check(load_library().WorldMi355Vibrato(self.handle, C.c_void_p(lf0.data_ptr()), off.ctypes.data_as(_ip),
                                                ss.ctypes.data_as(_ip), se.ctypes.data_as(_ip), sp_.ctypes.data_as(_dp),
                                                C.c_void_p(vib.data_ptr()), C.c_void_p(out.data_ptr()), C.byref(n)),
               "Vibrato")
        return vib, out, n.value

    def utterance_status(self, x=None, f0=None, sp=None, ap=None):
        """int32 cuda tensor [n_utt] of WM_UTT_* flags (1 non-finite input, 2 too short for Dio, 4 non-finite output)."""
        import torch
        st = torch.zeros(self.n_utt, dtype=torch.int32, device="cuda")
        ptr = lambda t: None if t is None else self._p(t)
        _check(load_library().WorldMi355UtteranceStatus(self.handle, ptr(x), ptr(f0), ptr(sp), ptr(ap),
                                                        C.c_void_p(st.data_ptr())), "UtteranceStatus")
        return st

    # ---- feature codec (world/codec.h) ----
    def code_spectral_envelope(self, sp, number_of_dimensions):
        import torch
        out = torch.empty(self.total_frames, number_of_dimensions, dtype=torch.float64, device="cuda")
        _check(load_library().WorldMi355CodeSpectralEnvelope(self.handle, self._p(sp), number_of_dimensions,
                                                             self._p(out)), "CodeSpectralEnvelope")
        return out

    def decode_spectral_envelope(self, coded):
        import torch
        out = torch.empty(self.total_frames, self.fft_size // 2 + 1, dtype=torch.float64, device="cuda")
        _check(load_library().WorldMi355DecodeSpectralEnvelope(self.handle, self._p(coded), int(coded.shape[1]),
                                                               self._p(out)), "DecodeSpectralEnvelope")
        return out

    def code_aperiodicity(self, ap):
        import torch
        nap = load_library().WorldMi355GetNumberOfAperiodicities(int(self.params.fs))
        out = torch.empty(self.total_frames, nap, dtype=torch.float64, device="cuda")
        _check(load_library().WorldMi355CodeAperiodicity(self.handle, self._p(ap), self._p(out)), "CodeAperiodicity")
        return out

    def decode_aperiodicity(self, coded):
        import torch
        out = torch.empty(self.total_frames, self.fft_size // 2 + 1, dtype=torch.float64, device="cuda")
        _check(load_library().WorldMi355DecodeAperiodicity(self.handle, self._p(coded), self._p(out)),
               "DecodeAperiodicity")
        return out

    def recipe_features(self, f0, sp, ap, spec_dim=50, ap_dim=25):
        """float32 lf0 / mgc / bap as the recipe's `analysis ... 5 2048 50 25` call writes them."""
        import torch
        lf0 = torch.empty(self.total_frames, dtype=torch.float32, device="cuda")
        mgc = torch.empty(self.total_frames, spec_dim, dtype=torch.float32, device="cuda")
        bap = torch.empty(self.total_frames, ap_dim, dtype=torch.float32, device="cuda")
        _check(load_library().WorldMi355RecipeFeatures(self.handle, self._p(f0), self._p(sp), self._p(ap), spec_dim,
                                                       ap_dim, C.c_void_p(lf0.data_ptr()), C.c_void_p(mgc.data_ptr()),
                                                       C.c_void_p(bap.data_ptr())), "RecipeFeatures")
        return lf0, mgc, bap

    def recipe_decode(self, lf0, mgc, bap):
        """f0 / sp / ap from the recipe's float32 lf0 / mgc / bap, as the synth CLI rebuilds them
        (test/synth.cpp:151-256); ap bins beyond the coding order are 0 (uninitialised in the reference)."""
        import torch
        for v in (lf0, mgc, bap):
            assert v.dtype == torch.float32 and v.is_cuda and v.is_contiguous()
        f0 = torch.empty(self.total_frames, dtype=torch.float64, device="cuda")
        sp = torch.empty(self.total_frames, self.bins, dtype=torch.float64, device="cuda")
        ap = torch.empty(self.total_frames, self.bins, dtype=torch.float64, device="cuda")
        _check(load_library().WorldMi355RecipeDecode(self.handle, C.c_void_p(lf0.data_ptr()), C.c_void_p(mgc.data_ptr()),
                                                     C.c_void_p(bap.data_ptr()), mgc.shape[1], bap.shape[1],
                                                     self._p(f0), self._p(sp), self._p(ap)), "RecipeDecode")
        return f0, sp, ap

    def compose_cmp(self, streams):
        """streams: list of (float32 cuda tensor [total_frames][dim], list of window coefficient lists).
        Returns float32 [total_frames][sum n_windows * dim] (window.pl + merge of the recipe's cmp stage)."""
        import torch
        n = len(streams)
        dp = C.POINTER(C.c_double)
        data = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) for t, _ in streams])
        dims = (C.c_int * n)(*[int(t.shape[1]) for t, _ in streams])
        nwin = (C.c_int * n)(*[len(w) for _, w in streams])
        keep, wptrs, sptrs = [], (C.POINTER(dp) * n)(), (C.POINTER(C.c_int) * n)()
        for s, (t, wins) in enumerate(streams):
            assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch.float32" and t.shape[0] == self.total_frames
            arrs = [(C.c_double * len(w))(*w) for w in wins]
            pa = (dp * len(wins))(*[C.cast(a, dp) for a in arrs])
            sz = (C.c_int * len(wins))(*[len(w) for w in wins])
            keep += [arrs, pa, sz]
            wptrs[s] = C.cast(pa, C.POINTER(dp))
            sptrs[s] = C.cast(sz, C.POINTER(C.c_int))
        cols = sum(int(t.shape[1]) * len(w) for t, w in streams)
        out = torch.empty(self.total_frames, cols, dtype=torch.float32, device="cuda")
        _check(load_library().WorldMi355ComposeCmp(self.handle, n, data, dims, nwin, wptrs, sptrs,
                                                   C.c_void_p(out.data_ptr())), "ComposeCmp")
        return out

    def split_frames(self, a):
        return [a[self.frame_offsets[u]:self.frame_offsets[u + 1]] for u in range(self.n_utt)]

    def split_out(self, y):
        return [y[self.out_offsets[u]:self.out_offsets[u + 1]] for u in range(self.n_utt)]

    def close(self):
        if self.handle:
            load_library().WorldMi355DestroyBatch(self.handle)
            self.handle = None


def write_files(items, threads=16):
    """items: [(path, numpy array)] -- every array written raw to its path by native threads in ONE library call
    (WorldMi355WriteFiles: the interpreter lock is released for its duration).  The arrays must be C-contiguous and
    stay alive until the call returns (they do: it is synchronous).  Host only."""
    n = len(items)
    if n == 0:
        return
    paths = (C.c_char_p * n)(*[os.fsencode(str(p)) for p, _ in items])
    ptrs = (C.c_void_p * n)()
    sizes = (C.c_size_t * n)()
    for k, (_, a) in enumerate(items):
        assert a.flags["C_CONTIGUOUS"], "write_files: contiguous arrays only"
        ptrs[k] = a.ctypes.data
        sizes[k] = a.nbytes
    _check(load_library().WorldMi355WriteFiles(n, paths, ptrs, sizes, int(threads)), "WriteFiles")


def htk_header(n_frames, sampling_rate, frame_shift_samples, bytes_per_frame, htk_type=9):
    """12-byte HTK header of the recipe's cmp files (addhtkheader.pl:60-75)."""
    buf = (C.c_ubyte * 12)()
    load_library().WorldMi355HtkHeader(int(n_frames), int(sampling_rate), int(frame_shift_samples), int(bytes_per_frame),
                                       int(htk_type), buf)
    return bytes(buf)
