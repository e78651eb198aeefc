"""Parity of the HIP path (through the C ABI of libworld_mi355.so) against the oracle.

Bars (BASELINE.json north_star): max|dF0| < 0.1 Hz, sp/ap RMSE < 1e-5 against the reference.
What is asserted here is far tighter (fp64 throughout, only the FFT factorisation and the
summation orders differ): |dF0| < 1e-6 Hz with identical V/UV, sp relative 1e-6, ap absolute
1e-8, y absolute 1e-8.
"""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
sd = importlib.import_module("hts-train-world_amd.synth_data")

F0_TOL, AP_TOL, Y_TOL = 1e-6, 1e-8, 1e-8


def sp_close(a, b):
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-13)


def oracle_chain(o, x, fs, fp=5.0, thr=0.0):
    t, f0d = o.dio(x, fs, fp)
    f0 = o.stonemask(x, fs, t, f0d)
    F = o.cheaptrick_fft_size(fs)
    sp = o.cheaptrick(x, fs, t, f0, -0.15, F)
    ap = o.d4c(x, fs, t, f0, F, thr)
    y = o.synthesis(f0, sp, ap, F, fp, fs)
    return dict(t=t, f0d=f0d, f0=f0, sp=sp, ap=ap, y=y, F=F)


def cat(rs, k):
    return np.concatenate([r[k] for r in rs])


def fft_hook(torch, x):
    """The product's wavefront FFT (csrc/fft.hpp) on the rows of x, through tests/hooks/libfft_hook.so."""
    import ctypes as C
    lib = C.CDLL(os.environ.get("WORLD_MI355_FFT_HOOK") or os.path.join(os.path.dirname(__file__), "hooks", "libfft_hook.so"))
    count, n = x.shape
    re = torch.empty(count, n // 2 + 1, dtype=torch.float64, device="cuda")
    im, xb = torch.empty_like(re), torch.empty_like(x)
    vp = C.c_void_p
    lib.FftHookRfft.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp]
    torch.cuda.synchronize()
    rc = lib.FftHookRfft(vp(torch.cuda.current_stream().cuda_stream), n, count, vp(x.data_ptr()), vp(re.data_ptr()),
                         vp(im.data_ptr()), vp(xb.data_ptr()))
    assert rc == 0
    return re, im, xb


@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_wavefront_fft_vs_numpy(gpu, n):
    torch, W, ctx = gpu
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(96, n, dtype=torch.float64, device="cuda", generator=g)
    re, im, xb = fft_hook(torch, x)
    ref = np.fft.rfft(x.cpu().numpy(), axis=1)
    scale = np.abs(ref).max()
    assert np.abs(re.cpu().numpy() + 1j * im.cpu().numpy() - ref).max() < 1e-14 * n * scale
    assert np.abs(xb.cpu().numpy() / n - x.cpu().numpy()).max() < 1e-13


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
def test_real_transforms_by_pairs(gpu, n):
    """rfft_split_pairs / rfft_backward_pairs (csrc/fft.hpp): the lane that holds Z[k] computes X[k] AND X[N - k], only the
    upper half of Z travels through LDS, and the inverse takes the pairs back.  Every bin against numpy, the round
    trip against the input, the pruned first pass (rows zero from some register on) against the plain one; 512 points
    ride on the 512-point complex plan (CheapTrick at fs <= 12.8 kHz)."""
    import ctypes as C
    torch, W, ctx = gpu
    lib = C.CDLL(os.environ.get("WORLD_MI355_FFT_HOOK") or os.path.join(os.path.dirname(__file__), "hooks", "libfft_hook.so"))
    vp = C.c_void_p
    lib.FftHookRfftPairs.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    g = torch.Generator(device="cuda").manual_seed(7 * n)
    M = n // 128
    for reach in (M, max(1, M // 4), max(1, M // 2)):
        x = torch.randn(64, n, dtype=torch.float64, device="cuda", generator=g)
        x[:, 128 * reach:] = 0.0
        ref = np.fft.rfft(x.cpu().numpy(), axis=1)
        scale = np.abs(ref).max()
        for nz in (-1, reach):
            re = torch.full((64, n // 2 + 1), np.nan, dtype=torch.float64, device="cuda")
            im, xb = torch.full_like(re, np.nan), torch.empty_like(x)
            torch.cuda.synchronize()
            assert lib.FftHookRfftPairs(vp(torch.cuda.current_stream().cuda_stream), n, 64, nz, vp(x.data_ptr()),
                                        vp(re.data_ptr()), vp(im.data_ptr()), vp(xb.data_ptr())) == 0
            got = re.cpu().numpy() + 1j * im.cpu().numpy()
            assert np.isfinite(got).all(), "a bin no lane wrote"
            assert np.abs(got - ref).max() < 1e-14 * n * scale, (reach, nz)
            assert np.abs(got[:, [0, -1]].imag).max() == 0.0          # DC and Nyquist: real, exactly
            assert np.abs(xb.cpu().numpy() / n - x.cpu().numpy()).max() < 1e-13, (reach, nz)


@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_pruned_first_pass_of_the_fft(gpu, n):
    """rfft_forward_nz (csrc/fft.hpp): the first pass of a transform whose operand is zero from some register on skips
    what the zeros make trivial (DftNz), in four sizes chosen at run time.  Rows of every length class, zero-padded,
    through the pruned pass declared with exactly their reach and with more than it: the spectrum of the plain
    transform to rounding (the skipped additions are of exact zeros: the same bits but for the sign of a zero)."""
    import ctypes as C
    torch, W, ctx = gpu
    lib = C.CDLL(os.environ.get("WORLD_MI355_FFT_HOOK") or os.path.join(os.path.dirname(__file__), "hooks", "libfft_hook.so"))
    vp = C.c_void_p
    lib.FftHookRfftNz.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    M = n // 128                                           # packed registers per lane
    g = torch.Generator(device="cuda").manual_seed(3 * n)
    for reach in (1, 2, M // 8, M // 4, M // 4 + 1, M // 2, M // 2 + 1, M - 1, M):
        reach = max(1, reach)
        x = torch.randn(8, n, dtype=torch.float64, device="cuda", generator=g)
        x[:, 128 * reach:] = 0.0
        x[3, 128 * reach - 5:] = 0.0                       # a window that ends inside its last register
        ref = np.fft.rfft(x.cpu().numpy(), axis=1)
        scale = np.abs(ref).max()
        outs = []
        for nz in (-1, reach, min(M, reach + 1), M):
            re = torch.empty(8, n // 2 + 1, dtype=torch.float64, device="cuda")
            im, xb = torch.empty_like(re), torch.empty_like(x)
            torch.cuda.synchronize()
            assert lib.FftHookRfftNz(vp(torch.cuda.current_stream().cuda_stream), n, 8, nz, vp(x.data_ptr()),
                                     vp(re.data_ptr()), vp(im.data_ptr()), vp(xb.data_ptr())) == 0
            got = re.cpu().numpy() + 1j * im.cpu().numpy()
            assert np.abs(got - ref).max() < 1e-14 * n * scale, (reach, nz)
            outs.append(got)
        for got in outs[1:]:
            assert np.abs(got - outs[0]).max() <= 1e-15 * n * scale


@pytest.fixture(scope="module")
def batch16(gpu, oracle):
    torch, W, ctx = gpu
    fs = 16000
    xs = [sd.make_utterance(i, fs, (2.0, 4.0)) for i in range(4)]
    rs = [oracle_chain(oracle, x, fs) for x in xs]
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    xc = torch.from_numpy(np.concatenate(xs)).cuda()
    yield torch, b, xs, xc, rs
    b.close()


def test_each_stage_against_oracle(batch16):
    torch, b, xs, xc, rs = batch16
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    t, f0d = b.dio(xc)
    np.testing.assert_array_equal(t.cpu().numpy(), cat(rs, "t"))
    assert ((f0d.cpu().numpy() > 0) == (cat(rs, "f0d") > 0)).all()
    np.testing.assert_allclose(f0d.cpu().numpy(), cat(rs, "f0d"), atol=F0_TOL, rtol=0)
    f0 = b.stonemask(xc, dev(cat(rs, "t")), dev(cat(rs, "f0d")))
    np.testing.assert_allclose(f0.cpu().numpy(), cat(rs, "f0"), atol=F0_TOL, rtol=0)
    sp = b.cheaptrick(xc, dev(cat(rs, "t")), dev(cat(rs, "f0")))
    sp_close(sp.cpu().numpy(), cat(rs, "sp"))
    ap = b.d4c(xc, dev(cat(rs, "t")), dev(cat(rs, "f0")))
    np.testing.assert_allclose(ap.cpu().numpy(), cat(rs, "ap"), atol=AP_TOL, rtol=0)
    y = b.synthesize(dev(cat(rs, "f0")), dev(cat(rs, "sp")), dev(cat(rs, "ap")))
    np.testing.assert_allclose(y.cpu().numpy(), cat(rs, "y"), atol=Y_TOL, rtol=0)


def test_pipeline_meets_the_north_star_bars(batch16):
    torch, b, xs, xc, rs = batch16
    t, f0, sp, ap = b.analyze(xc)
    y = b.synthesize(f0, sp, ap)
    f0, sp, ap, y = (v.cpu().numpy() for v in (f0, sp, ap, y))
    assert np.abs(f0 - cat(rs, "f0")).max() < 0.1
    assert np.sqrt(np.mean((sp - cat(rs, "sp")) ** 2)) < 1e-5
    assert np.sqrt(np.mean((ap - cat(rs, "ap")) ** 2)) < 1e-5
    # and the tight versions
    assert np.abs(f0 - cat(rs, "f0")).max() < F0_TOL
    sp_close(sp, cat(rs, "sp"))
    np.testing.assert_allclose(ap, cat(rs, "ap"), atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(y, cat(rs, "y"), atol=Y_TOL, rtol=0)


def test_golden_config1_on_gpu(gpu):
    """BASELINE config 1 shape against the vectors the compiled reference produced."""
    torch, W, ctx = gpu
    g = np.load(os.path.join(GOLDEN, "world_16k_cfg1.npz"))
    x = sd.make_utterance(int(g["index"]), 16000, duration=float(g["duration"]))
    b = W.WorldBatch(ctx, W.default_params(16000, 5.0), x_lengths=[len(x)])
    t, f0, sp, ap = b.analyze(torch.from_numpy(x).cuda())
    y = b.synthesize(f0, sp, ap)
    fs_, ss = int(g["frame_step"]), int(g["sample_step"])
    np.testing.assert_array_equal(t.cpu().numpy(), g["t"])
    np.testing.assert_allclose(f0.cpu().numpy(), g["f0"], atol=F0_TOL, rtol=0)
    sp_close(sp.cpu().numpy()[::fs_], g["sp_sub"])
    np.testing.assert_allclose(ap.cpu().numpy()[::fs_], g["ap_sub"], atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(y.cpu().numpy()[::ss], g["y_sub"], atol=Y_TOL, rtol=0)
    b.close()


def test_world_c_api_drop_in(pkg, oracle):
    """The reference's own entry points (host double*, double** rows) exported by the library."""
    C = pkg.capi
    fs = 16000
    x = sd.make_utterance(21, fs, duration=1.0)
    r = oracle_chain(oracle, x, fs, thr=0.85)
    t, f0d = C.dio(x, fs)
    np.testing.assert_array_equal(t, r["t"])
    np.testing.assert_allclose(f0d, r["f0d"], atol=F0_TOL, rtol=0)
    f0 = C.stonemask(x, fs, t, f0d)
    np.testing.assert_allclose(f0, r["f0"], atol=F0_TOL, rtol=0)
    assert C.cheaptrick_fft_size(fs) == r["F"]
    sp = C.cheaptrick(x, fs, t, f0)
    sp_close(sp, r["sp"])
    ap = C.d4c(x, fs, t, f0, r["F"])                # library default threshold 0.85
    np.testing.assert_allclose(ap, r["ap"], atol=AP_TOL, rtol=0)
    y = C.synthesis(f0, sp, ap, r["F"], 5.0, fs)
    np.testing.assert_allclose(y, r["y"], atol=Y_TOL, rtol=0)


def test_48k_path(gpu, oracle):
    """fft_size 2048, D4C at 4096, 5 aperiodicity bands."""
    torch, W, ctx = gpu
    fs = 48000
    x = sd.make_utterance(9, fs, duration=0.8)
    r = oracle_chain(oracle, x, fs)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    assert b.fft_size == 2048
    t, f0, sp, ap = b.analyze(torch.from_numpy(x).cuda())
    y = b.synthesize(f0, sp, ap)
    np.testing.assert_allclose(f0.cpu().numpy(), r["f0"], atol=F0_TOL, rtol=0)
    sp_close(sp.cpu().numpy(), r["sp"])
    np.testing.assert_allclose(ap.cpu().numpy(), r["ap"], atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(y.cpu().numpy(), r["y"], atol=Y_TOL, rtol=0)
    b.close()


@pytest.mark.parametrize("fs,fft", [(22050, 1024), (32000, 2048), (44100, 2048)])
def test_other_sampling_rates(gpu, oracle, fs, fft):
    """The rates between the two benchmark configurations: other FFT-size / band-count combinations
    (22.05 kHz: fft 1024, D4C 2048, 2 bands; 32 kHz: 2048 / 4096 / 4 bands; 44.1 kHz: 2048 / 4096 / 5 bands)."""
    torch, W, ctx = gpu
    x = sd.make_utterance(30 + fs // 1000, fs, duration=0.7)
    r = oracle_chain(oracle, x, fs)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    assert b.fft_size == fft == r["F"]
    t, f0, sp, ap = b.analyze(torch.from_numpy(x).cuda())
    y = b.synthesize(f0, sp, ap)
    assert ((f0.cpu().numpy() > 0) == (r["f0"] > 0)).all()
    np.testing.assert_allclose(f0.cpu().numpy(), r["f0"], atol=F0_TOL, rtol=0)
    sp_close(sp.cpu().numpy(), r["sp"])
    np.testing.assert_allclose(ap.cpu().numpy(), r["ap"], atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(y.cpu().numpy(), r["y"], atol=Y_TOL, rtol=0)
    th, fh = b.harvest(torch.from_numpy(x).cuda())
    to, fo = oracle.harvest(x, fs, 5.0)
    assert ((fh.cpu().numpy() > 0) == (fo > 0)).all()
    np.testing.assert_allclose(fh.cpu().numpy(), fo, atol=F0_TOL, rtol=0)
    b.close()


def test_pulse_kernel_forms_at_fft_2048(gpu, oracle):
    """Synthesis at fft 2048 renders its pulses with the spectra held by pairs in registers (csrc/synth_pulse_bp.hpp);
    WORLD_MI355_PULSE_BP=0 (read per launch) selects the strided form the other sizes use.  Same arithmetic per bin:
    both against the oracle at the tolerance of y, and against each other far below it."""
    torch, W, ctx = gpu
    fs = 48000
    xs = [sd.make_utterance(70 + i, fs, duration=0.5 + 0.1 * i) for i in range(3)]
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    assert b.fft_size == 2048
    t, f0, sp, ap = b.analyze(torch.from_numpy(np.concatenate(xs)).cuda())
    old = os.environ.get("WORLD_MI355_PULSE_BP")
    try:
        os.environ["WORLD_MI355_PULSE_BP"] = "1"
        y_pairs = b.synthesize(f0, sp, ap).cpu().numpy()
        os.environ["WORLD_MI355_PULSE_BP"] = "0"
        y_strided = b.synthesize(f0, sp, ap).cpu().numpy()
    finally:
        if old is None:
            os.environ.pop("WORLD_MI355_PULSE_BP", None)
        else:
            os.environ["WORLD_MI355_PULSE_BP"] = old
    assert np.abs(y_pairs - y_strided).max() < 1e-11
    r = oracle_chain(oracle, xs[0], fs)
    n = len(r["y"])
    np.testing.assert_allclose(y_pairs[:n], r["y"], atol=Y_TOL, rtol=0)
    np.testing.assert_allclose(y_strided[:n], r["y"], atol=Y_TOL, rtol=0)
    b.close()


_D4C_Q_CHILD = r"""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
pkg = importlib.import_module("hts-train-world_amd")
from oracle.bindings import Oracle
W, sd, o = pkg.world, pkg.synth_data, Oracle()
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
worst = 0.0
for fs, seeds, dur in ((16000, (3, 4, 90), 1.2), (22050, (52,), 0.7), (14000, (7,), 0.8)):
    xs = [sd.make_utterance(i, fs, duration=dur) for i in seeds]
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    t, f0, sp, ap = b.analyze(torch.from_numpy(np.concatenate(xs)).cuda())
    for u, x in enumerate(xs):
        g = slice(b.frame_offsets[u], b.frame_offsets[u + 1])
        to, f0d = o.dio(x, fs, 5.0)
        f0o = o.stonemask(x, fs, to, f0d)
        # low-pitched frames too (windows longer than a quarter of the transform): half of the contour an octave down
        apo = o.d4c(x, fs, to, f0o, b.fft_size, 0.0)
        worst = max(worst, float(np.abs(ap[g].cpu().numpy() - apo).max()))
    # the same frames an octave down: four periods then reach into the second quarter of the transform
    f0_low = torch.where(f0 > 0, f0 * 0.5, f0)
    ap_low = b.d4c(torch.from_numpy(np.concatenate(xs)).cuda(), t, f0_low)
    for u, x in enumerate(xs):
        g = slice(b.frame_offsets[u], b.frame_offsets[u + 1])
        apo = o.d4c(x, fs, t[g].cpu().numpy(), f0_low[g].cpu().numpy(), b.fft_size, 0.0)
        worst = max(worst, float(np.abs(ap_low[g].cpu().numpy() - apo).max()))
    b.close()
print("WORST", worst)
"""


def test_d4c_three_wave_form(gpu):
    """WORLD_MI355_D4C_Q=1: D4C at fft_size_d4c 2048 on the 512-point engine at three waves per SIMD (d4c_q.hpp; off by
    default).  The switch is read once per process, so a child process runs it: 16 kHz (one band),
    22.05 kHz (two bands), 14 kHz (a longer band window), each also an octave down (frames longer than a quarter of
    the transform), against the oracle."""
    env = dict(os.environ, WORLD_MI355_D4C_Q="1")
    out = subprocess.run([sys.executable, "-c", _D4C_Q_CHILD.format(root=ROOT)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    worst = float(out.stdout.strip().splitlines()[-1].split()[1])
    assert worst < AP_TOL, worst


def test_fft_size_512_at_8khz(gpu, pkg, oracle):
    """fs <= 12.8 kHz: GetFFTSizeForCheapTrick gives 512 (cheaptrick.cpp:191-194), D4C has no band at all
    (fs / 2 - 3000 < 3000, d4c.cpp:351-353).  Whole chain, codec and the drop-in entry points."""
    torch, W, ctx = gpu
    fs = 8000
    xs = [sd.make_utterance(110, fs, duration=0.8), sd.make_utterance(111, fs, duration=1.3)]
    rs = [oracle_chain(oracle, x, fs) for x in xs]
    assert rs[0]["F"] == 512
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    assert b.fft_size == 512
    t, f0, sp, ap, y = b.analyze_synthesize(torch.from_numpy(np.concatenate(xs)).cuda())
    assert ((f0.cpu().numpy() > 0) == (cat(rs, "f0") > 0)).all() and (cat(rs, "f0") > 0).sum() > 50
    np.testing.assert_allclose(f0.cpu().numpy(), cat(rs, "f0"), atol=F0_TOL, rtol=0)
    sp_close(sp.cpu().numpy(), cat(rs, "sp"))
    np.testing.assert_allclose(ap.cpu().numpy(), cat(rs, "ap"), atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(y.cpu().numpy(), cat(rs, "y"), atol=Y_TOL, rtol=0)
    # codec at 512: 20 coefficients (the reference reads spectrum[i] for i <= fft_size / 4)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    csp = b.code_spectral_envelope(dev(cat(rs, "sp")), 20).cpu().numpy()
    ref = oracle.code_spectral_envelope(cat(rs, "sp"), fs, 512, 20)
    np.testing.assert_allclose(csp, ref, atol=1e-11, rtol=0)
    np.testing.assert_allclose(b.decode_spectral_envelope(dev(ref)).cpu().numpy(),
                               oracle.decode_spectral_envelope(ref, fs, 512), rtol=1e-10)
    b.close()
    # drop-in entry points, one utterance
    C = pkg.capi
    r, x = rs[0], xs[0]
    sp1 = C.cheaptrick(x, fs, r["t"], r["f0"])
    sp_close(sp1, r["sp"])
    np.testing.assert_allclose(C.d4c(x, fs, r["t"], r["f0"], 512, threshold=0.0), r["ap"], atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(C.synthesis(r["f0"], r["sp"], r["ap"], 512, 5.0, fs), r["y"], atol=Y_TOL, rtol=0)


@pytest.mark.parametrize("fs", [96000, 88200])
def test_fft_size_4096_above_51khz(gpu, oracle, fs):
    """fs > 51.2 kHz (configure.ac:540-549 lists FFTLEN 4096 up to 102.4 kHz): CheapTrick / Synthesis / codec at
    fft_size 4096 and D4C with its own transform of 8192 points (d4c.cpp:344-346) -- the whole chain against the
    oracle, with the recipe's threshold 0 and with the API's default 0.85 (LoveTrain at 8192 points as well)."""
    torch, W, ctx = gpu
    x = sd.make_utterance(112, fs, duration=0.5)
    F = oracle.cheaptrick_fft_size(fs)
    assert F == 4096
    r = oracle_chain(oracle, x, fs)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    assert b.fft_size == 4096
    xd = torch.from_numpy(x).cuda()
    t, f0, sp, ap = b.analyze(xd)
    y = b.synthesize(f0, sp, ap)
    np.testing.assert_array_equal(t.cpu().numpy(), r["t"])
    assert ((f0.cpu().numpy() > 0) == (r["f0"] > 0)).all() and (r["f0"] > 0).sum() > 20
    np.testing.assert_allclose(f0.cpu().numpy(), r["f0"], atol=F0_TOL, rtol=0)
    sp_close(sp.cpu().numpy(), r["sp"])
    np.testing.assert_allclose(ap.cpu().numpy(), r["ap"], atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(y.cpu().numpy(), r["y"], atol=Y_TOL, rtol=0)
    b.close()
    # D4C alone with the default threshold of InitializeD4COption (d4c.cpp:399-401): LoveTrain decides per frame
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0, d4c_threshold=0.85), x_lengths=[len(x)])
    ap85 = b.d4c(xd, dev(r["t"]), dev(r["f0"])).cpu().numpy()
    want = oracle.d4c(x, fs, r["t"], r["f0"], F, 0.85)
    np.testing.assert_allclose(ap85, want, atol=AP_TOL, rtol=0)
    assert not np.array_equal(want, r["ap"])                          # the threshold did skip some frames
    # the codec at this size
    csp = b.code_spectral_envelope(sp, 60).cpu().numpy()
    ref = oracle.code_spectral_envelope(sp.cpu().numpy(), fs, F, 60)
    np.testing.assert_allclose(csp, ref, atol=1e-11, rtol=0)
    np.testing.assert_allclose(b.decode_spectral_envelope(dev(ref)).cpu().numpy(),
                               oracle.decode_spectral_envelope(ref, fs, F), rtol=1e-10)
    b.close()


@pytest.mark.parametrize("fs", [96000, 88200, 48000])
def test_d4c_above_fs_over_16_where_its_transform_has_8192_points(gpu, oracle, fs):
    """Frames with f0 >= fs / 16 (the smoothing mirrors reach past an eighth of the spectrum).  Up to 48 kHz the
    one-wavefront RARE kernel has taken them since round 2; where D4C's own transform has 8192 points (88.2 / 96 kHz)
    they kept the default row until round 5 -- the one place the library differed from the reference on valid input.
    d4c_wide_kernel (a workgroup per frame, direct DFTs) analyses them now: every frame is the oracle's, nothing is
    flagged, and the frames around them are untouched by the extra launch."""
    torch, W, ctx = gpu
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    x = sd.make_utterance(113, fs, duration=0.25)
    t = np.arange(int(1000.0 * len(x) / fs / 5.0) + 1) * 0.005
    f0 = np.full(len(t), 180.0)
    f0[5] = fs / 16.0 + 50.0
    f0[11] = fs / 16.0 - 50.0
    f0[17] = fs / 16.0                                   # the boundary itself
    f0[23] = fs / 5.0                                    # mirrors of 40 % of the spectrum
    f0[29] = fs / 2.0 - 3.0 * fs / 8192.0                # the widest the reference defines
    f0[31] = 0.0
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x), len(x)])
    xx, tt = dev(np.concatenate([x, x])), dev(np.concatenate([t, t]))
    f2 = np.concatenate([f0, np.full(len(t), 180.0)])           # the second utterance has no such frame
    ap = b.d4c(xx, tt, dev(f2))
    st = b.utterance_status(xx, dev(f2), None, ap).cpu().numpy()
    assert list(st) == [0, 0]
    got = b.split_frames(ap)
    want = oracle.d4c(x, fs, t, f0, b.fft_size, 0.0)
    for k in (5, 17, 23, 29):
        assert not np.all(want[k] == 1.0 - 1e-12)              # the reference does analyse them
    np.testing.assert_allclose(got[0].cpu().numpy(), want, atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(got[1].cpu().numpy(), oracle.d4c(x, fs, t, np.full(len(t), 180.0), b.fft_size, 0.0),
                               atol=AP_TOL, rtol=0)
    # with LoveTrain deciding (threshold 0.85): the same frames, those it lets through
    b85 = W.WorldBatch(ctx, W.default_params(fs, 5.0, d4c_threshold=0.85), x_lengths=[len(x)])
    np.testing.assert_allclose(b85.d4c(dev(x), dev(t), dev(f0)).cpu().numpy(), oracle.d4c(x, fs, t, f0, b.fft_size, 0.85),
                               atol=AP_TOL, rtol=0)
    b85.close()
    b.close()


@pytest.mark.parametrize("fs", [12500, 25000, 50000])
def test_rates_where_d4c_and_lovetrain_transforms_differ(gpu, oracle, fs):
    """fs in [12.0, 13.6), [24.1, 27.3), [48.1, 54.6) kHz: D4C's own transform is twice LoveTrain's (d4c.cpp:344-346 against
    :261-263).  Refused until round 5 (the two were required to be equal); the two kernels are sized independently."""
    torch, W, ctx = gpu
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    x = sd.make_utterance(70 + fs // 1000, fs, duration=0.6)
    r = oracle_chain(oracle, x, fs)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    t, f0, sp, ap = b.analyze(dev(x))
    y = b.synthesize(f0, sp, ap)
    assert ((f0.cpu().numpy() > 0) == (r["f0"] > 0)).all() and (r["f0"] > 0).sum() > 20
    np.testing.assert_allclose(f0.cpu().numpy(), r["f0"], atol=F0_TOL, rtol=0)
    sp_close(sp.cpu().numpy(), r["sp"])
    np.testing.assert_allclose(ap.cpu().numpy(), r["ap"], atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(y.cpu().numpy(), r["y"], atol=Y_TOL, rtol=0)
    b.close()
    if fs < 15800:
        # below 15.8 kHz D4CLoveTrainSub accumulates its power spectrum up to ceil(7900 fft / fs), past the Nyquist bin,
        # i.e. over memory it never wrote (d4c.cpp:243-247): with a threshold above zero the reference is undefined there
        return
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0, d4c_threshold=0.85), x_lengths=[len(x)])     # LoveTrain decides
    ap85 = b.d4c(dev(x), dev(r["t"]), dev(r["f0"])).cpu().numpy()
    np.testing.assert_allclose(ap85, oracle.d4c(x, fs, r["t"], r["f0"], r["F"], 0.85), atol=AP_TOL, rtol=0)
    b.close()


@pytest.mark.parametrize("fs,fp", [(16000, 1.0), (16000, 10.0), (16000, 2.5), (22050, 3.0), (48000, 4.0)])
def test_other_frame_periods(gpu, oracle, fs, fp):
    """Frame periods other than 5 ms, including ones that are not a whole number of samples (rounding ties)."""
    torch, W, ctx = gpu
    xs = [sd.make_utterance(50, fs, duration=0.6), sd.make_utterance(51, fs, duration=0.9)]
    rs = [oracle_chain(oracle, x, fs, fp) for x in xs]
    b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(x) for x in xs])
    t, f0, sp, ap = b.analyze(torch.from_numpy(np.concatenate(xs)).cuda())
    y = b.synthesize(f0, sp, ap)
    np.testing.assert_array_equal(t.cpu().numpy(), cat(rs, "t"))
    assert ((f0.cpu().numpy() > 0) == (cat(rs, "f0") > 0)).all()
    np.testing.assert_allclose(f0.cpu().numpy(), cat(rs, "f0"), atol=F0_TOL, rtol=0)
    sp_close(sp.cpu().numpy(), cat(rs, "sp"))
    np.testing.assert_allclose(ap.cpu().numpy(), cat(rs, "ap"), atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(y.cpu().numpy(), cat(rs, "y"), atol=Y_TOL, rtol=0)
    b.close()


def test_edge_cases(gpu, oracle):
    torch, W, ctx = gpu
    fs = 16000
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    # ragged batch: a 30 ms utterance (T = 7 <= voice_range_minimum -> zeros, dio.cpp:266) beside normal ones
    xs = [sd.make_utterance(30, fs, duration=0.03), sd.make_utterance(31, fs, duration=0.7),
          sd.make_utterance(32, fs, duration=0.031)]
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    t, f0 = b.dio(dev(np.concatenate(xs)))
    parts = b.split_frames(f0.cpu().numpy())
    assert len(parts[0]) == 7 and not parts[0].any() and not parts[2].any()
    to, fo = oracle.dio(xs[1], fs)
    np.testing.assert_allclose(parts[1], fo, atol=F0_TOL, rtol=0)
    # the one-call round trip on the same ragged batch (7-frame utterances have no voiced pulse at all): twice,
    # the first use of a batch takes the plain order, the second the two-stream one
    xr = dev(np.concatenate(xs))
    ta, f0a, spa, apa = b.analyze(xr)
    ya = b.synthesize(f0a, spa, apa)
    for _ in range(2):
        t2, f02, sp2, ap2, y2 = b.analyze_synthesize(xr)
        assert torch.equal(f0a, f02) and torch.equal(spa, sp2) and torch.equal(apa, ap2) and torch.equal(ya, y2)
    assert torch.isfinite(y2).all()
    b.close()
    # all-unvoiced features: CheapTrick at the 500 Hz default, D4C leaves 1 - 1e-12, Synthesis is noise only
    x = xs[1]
    nf = len(to)
    z = np.zeros(nf)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    sp = b.cheaptrick(dev(x), dev(to), dev(z))
    sp_close(sp.cpu().numpy(), oracle.cheaptrick(x, fs, to, z))
    ap = b.d4c(dev(x), dev(to), dev(z)).cpu().numpy()
    assert np.all(ap == 1.0 - 1e-12)
    y = b.synthesize(dev(z), sp, dev(ap))
    np.testing.assert_allclose(y.cpu().numpy(), oracle.synthesis(z, sp.cpu().numpy(), ap, 1024, 5.0, fs), atol=Y_TOL)
    # StoneMask gates: f0 <= 40 or > fs/12 -> 0 (stonemask.cpp:186-187)
    weird = np.full(nf, 200.0)
    weird[::3] = 39.0
    weird[1::3] = fs / 12.0 + 1.0
    sm = b.stonemask(dev(x), dev(to), dev(weird)).cpu().numpy()
    np.testing.assert_allclose(sm, oracle.stonemask(x, fs, to, weird), atol=F0_TOL, rtol=0)
    assert not sm[::3].any() and not sm[1::3].any()
    b.close()
    # D4C at the library default threshold 0.85 (frames failing LoveTrain keep 1 - 1e-12)
    p = W.default_params(fs, 5.0, d4c_threshold=0.85)
    b = W.WorldBatch(ctx, p, x_lengths=[len(x)])
    f0r = oracle.stonemask(x, fs, to, fo)
    ap = b.d4c(dev(x), dev(to), dev(f0r)).cpu().numpy()
    np.testing.assert_allclose(ap, oracle.d4c(x, fs, to, f0r, 1024, 0.85), atol=AP_TOL, rtol=0)
    b.close()


@pytest.mark.parametrize("fs", [16000, 48000])
def test_d4c_outside_the_usual_f0_range(gpu, pkg, oracle, fs):
    """D4C analyses every frame with f0 != 0 (d4c.cpp:380-382), also where Dio would never put one: windows longer
    than half the D4C FFT (f0 towards the 47 Hz floor and below it) and smoothing widths beyond fs/16 (raised
    f0_ceil).  Those frames take the second instantiation of the kernel; the usual ones beside them the first."""
    torch, W, ctx = gpu
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    x = sd.make_utterance(70, fs, duration=0.6)
    t, f0d = oracle.dio(x, fs)
    nf = len(t)
    f0 = np.full(nf, 180.0)
    f0[0::7] = 30.0                  # below the floor: analysed at 47 Hz, window of 4 periods > FD / 2 samples
    f0[1::7] = 55.0                  # long window
    f0[2::7] = fs / 16.0 + 50.0      # mirror wider than FD / 16 bins
    f0[3::7] = 1200.0                # raised ceiling
    f0[4::7] = fs / 5.0
    f0[5::7] = 0.0
    F = oracle.cheaptrick_fft_size(fs)
    want = oracle.d4c(x, fs, t, f0, F, 0.0)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    got = b.d4c(dev(x), dev(t), dev(f0)).cpu().numpy()
    b.close()
    np.testing.assert_allclose(got, want, atol=AP_TOL, rtol=0)
    assert np.all(got[5::7] == 1.0 - 1e-12) and np.all(got[2::7] != 1.0 - 1e-12)
    # the same through the drop-in entry point, as a caller with a raised f0_ceil would reach it
    got2 = pkg.capi.d4c(x, fs, t, f0, F, threshold=0.0)
    np.testing.assert_array_equal(got2, got)


@pytest.mark.gpu
@pytest.mark.parametrize("fs", [16000, 48000])
def test_cheaptrick_outside_the_usual_f0_range(gpu, pkg, oracle, fs):
    """CheapTrick takes any f0 the caller hands it (cheaptrick.cpp:217 only replaces values at or below its floor).
    Frames whose smoothing mirror fits fft/8 bins run with the small margins, the others (f0 of a few kHz: a raised
    f0_ceil, or another estimator's output) with the full ones; both beside each other in one call."""
    torch, W, ctx = gpu
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    x = sd.make_utterance(71, fs, duration=0.6)
    t, _ = oracle.dio(x, fs)
    nf = len(t)
    F = oracle.cheaptrick_fft_size(fs)
    edge = (F // 8 - 2) * fs / F               # the largest f0 of the usual instantiation
    f0 = np.full(nf, 200.0)
    f0[0::6] = edge - 1.0
    f0[1::6] = edge + 1.0
    f0[2::6] = fs / 4.0
    f0[3::6] = fs / 3.0
    f0[4::6] = 0.0
    f0[5::12] = 0.45 * fs              # above fs / 3 DCCorrection touches more bins than the smoothing mirror is wide
    f0[11::12] = 0.4999 * fs           # the last f0 the reference defines (upper_limit = fft_size / 2 + 1, common.cpp:56-75)
    want = oracle.cheaptrick(x, fs, t, f0)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    got = b.cheaptrick(dev(x), dev(t), dev(f0)).cpu().numpy()
    b.close()
    sp_close(got, want)
    got2 = pkg.capi.cheaptrick(x, fs, t, f0)
    np.testing.assert_array_equal(got2, got)
    # beyond fs / 2 (undefined in the reference: DCCorrection runs past its spectrum) and non-finite values are
    # analysed at the default f0 like the values at or below the floor; nothing faults, every row is finite
    f0b = f0.copy()
    f0b[0::6], f0b[1::6], f0b[2::6] = fs * 0.75, np.inf, np.nan
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    gotb = b.cheaptrick(dev(x), dev(t), dev(f0b)).cpu().numpy()
    b.close()
    assert np.isfinite(gotb).all()
    f0z = f0b.copy()
    f0z[0::6], f0z[1::6], f0z[2::6] = 0.0, 0.0, 0.0
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    gotz = b.cheaptrick(dev(x), dev(t), dev(f0z)).cpu().numpy()
    b.close()
    np.testing.assert_array_equal(gotb, gotz)


@pytest.mark.parametrize("thr", [0.0, -1.0])
def test_d4c_threshold_zero_with_bad_samples(gpu, oracle, thr):
    """With a threshold <= 0 D4CLoveTrain cannot skip a voiced frame (its ratio is >= 0, zero is ruled out by the
    window's dither, and a NaN ratio does not satisfy `<= threshold`, d4c.cpp:380), so the library does not run its
    transform (d4c_lovetrain_all_pass_kernel).  Same rows as the reference's arithmetic, also around NaN / Inf
    samples, where that ratio is NaN and the rows are NaN as well."""
    torch, W, ctx = gpu
    fs = 16000
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    x = sd.make_utterance(95, fs, duration=0.8).copy()
    x[4000] = np.nan
    x[8000:8003] = np.inf
    x[11000] = -np.inf
    t, _ = oracle.dio(np.nan_to_num(x, posinf=0.0, neginf=0.0), fs)
    f0 = np.full(len(t), 170.0)
    f0[::9] = 0.0
    f0[3::9] = 45.0                  # LoveTrain's 3-period window at max(f0, 40) is then wider than the body's frames
    want = oracle.d4c(x, fs, t, f0, 1024, thr)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0, d4c_threshold=thr), x_lengths=[len(x)])
    got = b.d4c(dev(x), dev(t), dev(f0)).cpu().numpy()
    b.close()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.isnan(want).any() and (want == 1.0 - 1e-12).all(axis=1).sum() >= len(t) // 9     # both kinds of rows occur
    np.testing.assert_allclose(got, want, atol=AP_TOL, rtol=0, equal_nan=True)


def test_long_utterance_at_1ms_hop(gpu, oracle):
    """21 s at a 1 ms hop = 21 001 frames: the contour fix's edge lists (2 x (T / 2 + 2) ints) no longer fit the
    48 KB of LDS they use for ordinary lengths and move to global memory (dio.hip, dio_fix_kernel<false>)."""
    torch, W, ctx = gpu
    fs, fp = 16000, 1.0
    x = np.concatenate([sd.make_utterance(90 + k, fs, duration=7.0) for k in range(3)])
    b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(x)])
    assert b.total_frames == 21001
    t, f0 = b.dio(torch.from_numpy(x).cuda())
    to, fo = oracle.dio(x, fs, fp)
    np.testing.assert_array_equal(t.cpu().numpy(), to)
    np.testing.assert_allclose(f0.cpu().numpy(), fo, atol=F0_TOL, rtol=0)
    assert (fo > 0).sum() > 5000
    b.close()


def test_status_flags_and_batch_isolation(gpu):
    """A bad utterance is reported and does not touch its neighbours: NaN / Inf samples in
    one utterance of a batch, a too-short one beside it; the others come out bit-identical to a clean batch."""
    torch, W, ctx = gpu
    fs = 16000
    xs = [sd.make_utterance(80 + k, fs, duration=d) for k, d in enumerate((0.6, 0.9, 0.03, 0.7))]
    bad = [x.copy() for x in xs]
    bad[1][3000] = np.nan
    bad[1][7000:7004] = np.inf
    bad[1][9000] = -np.inf
    lens = [len(x) for x in xs]
    outs = []
    for batch_xs in (xs, bad):
        b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=lens)
        xc = torch.from_numpy(np.concatenate(batch_xs)).cuda()
        t, f0, sp, ap, y = b.analyze_synthesize(xc)
        st = b.utterance_status(xc, f0, sp, ap).cpu().numpy()
        outs.append((b.split_frames(f0), b.split_frames(sp), b.split_frames(ap), b.split_out(y), st))
        assert torch.isfinite(y[b.out_offsets[3]:]).all()
        b.close()
    clean, dirty = outs
    assert list(clean[4]) == [0, 0, 2, 0]                         # WM_UTT_TOO_SHORT: 7 frames
    assert dirty[4][1] & 1 and dirty[4][1] & 4 and list(dirty[4][[0, 2, 3]]) == [0, 2, 0]
    for u in (0, 2, 3):
        for k in range(4):
            assert torch.equal(clean[k][u], dirty[k][u]), (u, k)
    # (the damaged utterance itself is lost as a whole: Dio subtracts the mean over all its samples, dio.cpp:74-79)


def test_full_size_batches(gpu, oracle):
    """BASELINE.json's full sizes in the test suite proper: configs[1] (256 utterances of 2-8 s, analysis + synthesis)
    and configs[4] (Synthesis only over 1024 feature sets): properties that do not need the oracle on everything --
    run-to-run bit identity, ranges, the response scratch in several chunks giving the same bits -- plus spot
    parity on the shortest and the longest utterance."""
    torch, W, ctx = gpu
    fs = 16000
    xs = sd.make_batch(256, fs, (2.0, 8.0), first=0, workers=16)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    assert b.total_frames > 250_000
    xc = torch.from_numpy(np.concatenate(xs)).cuda()
    t, f0, sp, ap, y = b.analyze_synthesize(xc)
    r1 = [v.clone() for v in (f0, sp, ap, y)]
    t, f0, sp, ap, y = b.analyze_synthesize(xc)
    for a, c in zip(r1, (f0, sp, ap, y)):
        assert torch.equal(a, c)
    assert int(b.utterance_status(xc, f0, sp, ap).max()) == 0
    assert bool(((f0 == 0) | ((f0 >= 40.0) & (f0 <= 1000.0))).all())
    assert bool((sp > 0).all()) and bool(((ap > 0) & (ap <= 1.0)).all()) and float(y.abs().max()) < 2.0
    order = np.argsort([len(x) for x in xs])
    for u in (int(order[0]), int(order[-1])):
        r = oracle_chain(oracle, xs[u], fs)
        g = slice(b.frame_offsets[u], b.frame_offsets[u + 1])
        np.testing.assert_allclose(f0[g].cpu().numpy(), r["f0"], atol=F0_TOL, rtol=0)
        sp_close(sp[g].cpu().numpy(), r["sp"])
        np.testing.assert_allclose(ap[g].cpu().numpy(), r["ap"], atol=AP_TOL, rtol=0)
        np.testing.assert_allclose(y[b.out_offsets[u]:b.out_offsets[u + 1]].cpu().numpy(), r["y"], atol=Y_TOL, rtol=0)
    # configs[4]: 1024 feature sets = these 256 four times over, synthesised as one batch
    T = np.diff(b.frame_offsets).tolist() * 4
    Y = np.diff(b.out_offsets).tolist() * 4
    bs = W.WorldBatch(ctx, W.default_params(fs, 5.0), f0_lengths=T, y_lengths=Y)
    y4 = bs.synthesize(f0.repeat(4), sp.repeat(4, 1), ap.repeat(4, 1))
    n = int(b.total_out)
    for k in range(4):
        assert torch.equal(y4[k * n:(k + 1) * n], y)           # the same pulses whatever the batch around them
    bs.close()
    b.close()


def test_full_size_properties(gpu, oracle):
    """Config-2-sized durations (2-8 s): size-independent properties + spot parity."""
    torch, W, ctx = gpu
    fs = 16000
    xs = sd.make_batch(48, fs, (2.0, 8.0), first=100, workers=8)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    xc = torch.from_numpy(np.concatenate(xs)).cuda()
    t, f0, sp, ap = b.analyze(xc)
    y = b.synthesize(f0, sp, ap)
    r1 = [v.clone() for v in (f0, sp, ap, y)]
    t, f0, sp, ap = b.analyze(xc)
    y = b.synthesize(f0, sp, ap)
    for a, c in zip(r1, (f0, sp, ap, y)):
        assert torch.equal(a, c)                       # run-to-run bit-identical
    f0n, spn, apn, yn = (v.cpu().numpy() for v in (f0, sp, ap, y))
    assert np.all((f0n == 0) | ((f0n >= 40.0) & (f0n <= 1000.0)))
    assert np.all(spn > 0) and np.all(np.isfinite(spn))
    assert np.all((apn > 0) & (apn <= 1.0))
    assert np.all(np.isfinite(yn)) and np.abs(yn).max() < 2.0
    # the fused call (Synthesis' first part on a second stream) gives the same bits as the two calls
    t2, f02, sp2, ap2, y2 = b.analyze_synthesize(xc)
    for a, c in zip(r1, (f02, sp2, ap2, y2)):
        assert torch.equal(a, c)
    # batch invariance: an utterance analysed alone gives the same bits as inside the batch
    u = 17
    b1 = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(xs[u])])
    t1, f01, sp1, ap1 = b1.analyze(torch.from_numpy(xs[u]).cuda())
    g = slice(b.frame_offsets[u], b.frame_offsets[u + 1])
    assert torch.equal(f01, f0[g]) and torch.equal(sp1, sp[g]) and torch.equal(ap1, ap[g])
    b1.close()
    # spot parity on the longest and the shortest utterance
    order = np.argsort([len(x) for x in xs])
    for u in (order[0], order[-1]):
        r = oracle_chain(oracle, xs[u], fs)
        g = slice(b.frame_offsets[u], b.frame_offsets[u + 1])
        np.testing.assert_allclose(f0n[g], r["f0"], atol=F0_TOL, rtol=0)
        sp_close(spn[g], r["sp"])
        np.testing.assert_allclose(apn[g], r["ap"], atol=AP_TOL, rtol=0)
        np.testing.assert_allclose(yn[b.out_offsets[u]:b.out_offsets[u + 1]], r["y"], atol=Y_TOL, rtol=0)
    b.close()


@pytest.mark.parametrize("fs,fp,idxs,dur", [(16000, 5.0, [3, 6], (1.0, 2.0)), (48000, 1.0, [4, 8], (0.8, 1.6)),
                                             (22050, 5.0, [7], (1.2, 1.2)), (8000, 5.0, [14], (1.5, 1.5))])
def test_harvest_against_oracle(gpu, oracle, fs, fp, idxs, dur):
    """BASELINE config 3 shape (Harvest, 48 kHz, 1 ms) and friends; parity on f0 only."""
    torch, W, ctx = gpu
    xs = [sd.make_utterance(i, fs, dur) for i in idxs]
    b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(x) for x in xs])
    t, f0 = b.harvest(torch.from_numpy(np.concatenate(xs)).cuda())
    ro = [oracle.harvest(x, fs, fp) for x in xs]
    np.testing.assert_array_equal(t.cpu().numpy(), np.concatenate([r[0] for r in ro]))
    fo = np.concatenate([r[1] for r in ro])
    f0 = f0.cpu().numpy()
    assert ((f0 > 0) == (fo > 0)).all()
    assert np.abs(f0 - fo).max() < 0.1          # the bar
    np.testing.assert_allclose(f0, fo, atol=F0_TOL, rtol=0)
    b.close()


@pytest.mark.parametrize("fs,fp,lo,hi", [(16000, 5.0, 50.0, 500.0), (16000, 5.0, 100.0, 1000.0), (48000, 1.0, 50.0, 500.0),
                                         (22050, 2.0, 110.0, 420.0)])
def test_harvest_option_sweep(gpu, pkg, oracle, fs, fp, lo, hi):
    """HarvestOption.f0_floor / f0_ceil away from the defaults (harvest.cpp:1223-1262): the range sets the number of
    filter-bank channels (harvest.cpp:1139-1148: 1 + int(log2(ceil * 1.1 / (floor * 0.9)) * 40)) and the candidate
    capacity per frame (:334-343), which size the detection and refinement kernels; through the batched entry and
    through the drop-in Harvest()."""
    torch, W, ctx = gpu
    xs = [sd.make_utterance(i, fs, duration=d) for i, d in ((21, 0.9), (22, 1.4))]
    b = W.WorldBatch(ctx, W.default_params(fs, fp, f0_floor=lo, f0_ceil=hi), x_lengths=[len(x) for x in xs])
    t, f0 = b.harvest(torch.from_numpy(np.concatenate(xs)).cuda())
    b.close()
    ro = [oracle.harvest(x, fs, fp, lo, hi) for x in xs]
    np.testing.assert_array_equal(t.cpu().numpy(), np.concatenate([r[0] for r in ro]))
    fo = np.concatenate([r[1] for r in ro])
    f0 = f0.cpu().numpy()
    assert (fo > 0).sum() > len(fo) // 10               # the case is not trivially unvoiced
    assert ((f0 > 0) == (fo > 0)).all()
    np.testing.assert_allclose(f0, fo, atol=F0_TOL, rtol=0)
    t1, f1 = pkg.capi.harvest(xs[0], fs, fp, lo, hi)
    np.testing.assert_array_equal(t1, ro[0][0])
    np.testing.assert_allclose(f1, ro[0][1], atol=F0_TOL, rtol=0)


def test_harvest_full_size_config2(gpu, oracle):
    """BASELINE.json configs[2] at its full size: Harvest over 64 utterances of 2-8 s at 48 kHz with a 1 ms hop
    (about 320 k frames).  Properties that need no oracle on everything -- run-to-run bit identity, the time axis,
    the f0 range, batch invariance of one utterance -- plus
    parity on the shortest utterance against the oracle."""
    torch, W, ctx = gpu
    fs, fp = 48000, 1.0
    xs = sd.make_batch(64, fs, (2.0, 8.0), first=0, workers=16)
    b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(x) for x in xs])
    assert b.total_frames > 250_000
    xc = torch.from_numpy(np.concatenate(xs)).cuda()
    t, f0 = b.harvest(xc)
    t1, f01 = t.clone(), f0.clone()
    t, f0 = b.harvest(xc)
    assert torch.equal(t, t1) and torch.equal(f0, f01)
    fo = b.frame_offsets
    for u in (0, 31, 63):
        n = fo[u + 1] - fo[u]
        np.testing.assert_array_equal(t[fo[u]:fo[u + 1]].cpu().numpy(), np.arange(n) * fp / 1000.0)
    # refined candidates lie in [floor, ceil] (harvest.cpp:610-614); the zero-lag smoothing of the contour
    # (:1079-1113) may carry a value slightly past either end, nothing clips it afterwards
    assert bool(torch.isfinite(f0).all()) and bool(((f0 == 0) | ((f0 >= 50.0) & (f0 <= 1000.0))).all())
    voiced = float((f0 > 0).double().mean())
    assert 0.2 < voiced < 0.98
    order = np.argsort([len(x) for x in xs])
    u = int(order[0])
    b1 = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(xs[u])])
    _, f0a = b1.harvest(torch.from_numpy(xs[u]).cuda())
    b1.close()
    assert torch.equal(f0a, f0[fo[u]:fo[u + 1]])                    # the same bits alone and inside the batch
    _, want = oracle.harvest(xs[u], fs, fp)
    got = f0[fo[u]:fo[u + 1]].cpu().numpy()
    assert ((got > 0) == (want > 0)).all()
    np.testing.assert_allclose(got, want, atol=F0_TOL, rtol=0)
    b.close()


def test_harvest_golden_and_c_api(pkg):
    g = np.load(os.path.join(GOLDEN, "harvest_48k_1ms.npz"))
    x = sd.make_utterance(int(g["index"]), int(g["fs"]), duration=float(g["duration"]))
    t, f0 = pkg.capi.harvest(x, int(g["fs"]), float(g["frame_period"]))
    np.testing.assert_array_equal(t, g["t"])
    np.testing.assert_allclose(f0, g["f0"], atol=F0_TOL, rtol=0)
    g = np.load(os.path.join(GOLDEN, "harvest_16k.npz"))
    x = sd.make_utterance(int(g["index"]), int(g["fs"]), duration=float(g["duration"]))
    t, f0 = pkg.capi.harvest(x, int(g["fs"]), float(g["frame_period"]))
    np.testing.assert_allclose(f0, g["f0"], atol=F0_TOL, rtol=0)


@pytest.mark.parametrize("kw", [dict(speed=2), dict(speed=4), dict(speed=11), dict(speed=12, frame_period=10.0),
                                dict(f0_floor=50.0, f0_ceil=600.0), dict(channels_in_octave=4.0, allowed_range=0.05),
                                dict(frame_period=1.0)])
def test_dio_option_sweep(gpu, oracle, kw):
    """DioOption fields (dio.h:16-23), including the decimated speeds (dio.cpp:68-70, :589-591)."""
    torch, W, ctx = gpu
    fs = 16000 if kw.get("speed", 1) < 8 else 48000
    xs = [sd.make_utterance(40, fs, duration=0.9), sd.make_utterance(41, fs, duration=1.3)]
    fp = kw.get("frame_period", 5.0)
    p = W.default_params(fs, fp, **{k: v for k, v in kw.items() if k != "frame_period"})
    b = W.WorldBatch(ctx, p, x_lengths=[len(x) for x in xs])
    t, f0 = b.dio(torch.from_numpy(np.concatenate(xs)).cuda())
    ro = [oracle.dio(x, fs, fp, kw.get("f0_floor", 71.0), kw.get("f0_ceil", 800.0),
                     kw.get("channels_in_octave", 2.0), kw.get("speed", 1), kw.get("allowed_range", 0.1)) for x in xs]
    np.testing.assert_array_equal(t.cpu().numpy(), np.concatenate([r[0] for r in ro]))
    fo = np.concatenate([r[1] for r in ro])
    assert ((f0.cpu().numpy() > 0) == (fo > 0)).all()
    np.testing.assert_allclose(f0.cpu().numpy(), fo, atol=F0_TOL, rtol=0)
    b.close()


def test_fir_tile_edges(gpu, oracle):
    """DIO / Harvest filter tiles are 2046 samples; StoneMask, the synthesis chain and the smoothing scan work in
    chunks of 64..2048.  Lengths sitting on, just below and just above those boundaries, in one ragged batch."""
    torch, W, ctx = gpu
    fs = 16000
    base = sd.make_utterance(77, fs, duration=1.2)
    lens = [2045, 2046, 2047, 2 * 2046 - 1, 2 * 2046, 2 * 2046 + 1, 4 * 2046 + 1, 8191, 8192, 8193]
    xs = [np.ascontiguousarray(base[:n]) for n in lens]
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=lens)
    xc = torch.from_numpy(np.concatenate(xs)).cuda()
    t, f0d = b.dio(xc)
    ro = [oracle.dio(x, fs, 5.0) for x in xs]
    np.testing.assert_array_equal(t.cpu().numpy(), np.concatenate([r[0] for r in ro]))
    fo = np.concatenate([r[1] for r in ro])
    assert ((f0d.cpu().numpy() > 0) == (fo > 0)).all()
    np.testing.assert_allclose(f0d.cpu().numpy(), fo, atol=F0_TOL, rtol=0)
    th, f0h = b.harvest(xc)
    rh = [oracle.harvest(x, fs, 5.0) for x in xs]
    fh = np.concatenate([r[1] for r in rh])
    assert ((f0h.cpu().numpy() > 0) == (fh > 0)).all()
    np.testing.assert_allclose(f0h.cpu().numpy(), fh, atol=F0_TOL, rtol=0)
    b.close()
    # the sequential phase chain of Synthesis: output lengths around its 64 / 1024 / 2048 sample blocks
    x = sd.make_utterance(78, fs, duration=0.6)
    r = oracle_chain(oracle, x, fs)
    for ylen in (1023, 1024, 1025, 2047, 2048, 2049, 4097):
        yo = oracle.synthesis(r["f0"], r["sp"], r["ap"], r["F"], 5.0, fs, y_length=ylen)
        bb = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)], y_lengths=[ylen])
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        y = bb.synthesize(dev(r["f0"]), dev(r["sp"]), dev(r["ap"]))
        np.testing.assert_allclose(y.cpu().numpy(), yo, atol=Y_TOL, rtol=0)
        bb.close()


@pytest.mark.parametrize("fs,nd", [(16000, 50), (48000, 60)])
def test_codec_against_oracle(gpu, pkg, oracle, fs, nd):
    """world/codec.h on the device: batched API, the C ABI, the recipe packing."""
    torch, W, ctx = gpu
    xs = [sd.make_utterance(i, fs, duration=d) for i, d in ((21, 0.7), (22, 1.1))]
    rs = [oracle_chain(oracle, x, fs) for x in xs]
    F = rs[0]["F"]
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    sp, ap, f0 = cat(rs, "sp"), cat(rs, "ap"), cat(rs, "f0")
    csp = b.code_spectral_envelope(dev(sp), nd).cpu().numpy()
    ref_csp = oracle.code_spectral_envelope(sp, fs, F, nd)
    np.testing.assert_allclose(csp, ref_csp, atol=1e-11, rtol=0)
    dsp = b.decode_spectral_envelope(dev(ref_csp)).cpu().numpy()
    np.testing.assert_allclose(dsp, oracle.decode_spectral_envelope(ref_csp, fs, F), rtol=1e-10)
    cap = b.code_aperiodicity(dev(ap)).cpu().numpy()
    ref_cap = oracle.code_aperiodicity(ap, fs, F)
    np.testing.assert_allclose(cap, ref_cap, atol=1e-10, rtol=0)
    mix = ref_cap.copy()
    mix[::3] = -0.1                                             # CheckVUV keeps the default row
    dap = b.decode_aperiodicity(dev(mix)).cpu().numpy()
    np.testing.assert_allclose(dap, oracle.decode_aperiodicity(mix, fs, F), atol=1e-12, rtol=0)
    # recipe packing (float32): lf0 / mgc / bap
    from test_golden import recipe_pack
    lf0, mgc, bap = b.recipe_features(dev(f0), dev(sp), dev(ap), nd, 25)
    rl, rm, rb = recipe_pack(oracle, f0, sp, ap, fs, F, nd)
    np.testing.assert_allclose(lf0.cpu().numpy(), rl, atol=1e-6, rtol=0)
    np.testing.assert_allclose(mgc.cpu().numpy(), rm, atol=2e-6, rtol=0)
    np.testing.assert_allclose(bap.cpu().numpy(), rb, atol=2e-6, rtol=0)
    b.close()
    # C ABI, one utterance, host pointers
    C = pkg.capi
    assert C.get_number_of_aperiodicities(fs) == oracle.num_aperiodicities(fs)
    r = rs[0]
    np.testing.assert_allclose(C.code_spectral_envelope(r["sp"], fs, F, nd),
                               oracle.code_spectral_envelope(r["sp"], fs, F, nd), atol=1e-11, rtol=0)
    c1 = oracle.code_spectral_envelope(r["sp"], fs, F, nd)
    np.testing.assert_allclose(C.decode_spectral_envelope(c1, fs, F), oracle.decode_spectral_envelope(c1, fs, F), rtol=1e-10)
    a1 = oracle.code_aperiodicity(r["ap"], fs, F)
    np.testing.assert_allclose(C.code_aperiodicity(r["ap"], fs, F), a1, atol=1e-10, rtol=0)
    np.testing.assert_allclose(C.decode_aperiodicity(a1, fs, F), oracle.decode_aperiodicity(a1, fs, F), atol=1e-12, rtol=0)


def test_codec_golden_on_gpu(gpu, oracle):
    """The reference's own coded features (tests/golden/codec_16k.npz) from the device path end to end."""
    torch, W, ctx = gpu
    g = np.load(os.path.join(GOLDEN, "codec_16k.npz"))
    fs, F, nd = int(g["fs"]), int(g["fft_size"]), int(g["spec_dim"])
    x = sd.make_utterance(int(g["index"]), fs, duration=float(g["duration"]))
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    t, f0, sp, ap = b.analyze(torch.from_numpy(x).cuda())
    np.testing.assert_allclose(f0.cpu().numpy(), g["f0"], atol=F0_TOL, rtol=0)
    np.testing.assert_allclose(b.code_spectral_envelope(sp, nd).cpu().numpy(), g["coded_sp"], atol=1e-7, rtol=0)
    np.testing.assert_allclose(b.code_aperiodicity(ap).cpu().numpy(), g["coded_ap"], atol=1e-6, rtol=0)
    lf0, mgc, bap = b.recipe_features(f0, sp, ap, nd, 25)
    np.testing.assert_allclose(lf0.cpu().numpy(), g["lf0"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(mgc.cpu().numpy(), g["mgc"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(bap.cpu().numpy(), g["bap"], atol=1e-5, rtol=0)
    b.close()


def test_cmp_composition(gpu, pkg, oracle):
    """window.pl + merge + HTK header on the device, bit for bit against the reference
    scripts' outputs (golden) and, on a ragged two-utterance batch, against the oracle per utterance."""
    from test_golden import CMP_WINDOWS, cmp_stream
    torch, W, ctx = gpu
    g = np.load(os.path.join(GOLDEN, "cmp_windows.npz"))
    fs = 16000
    T = int(g["mgc_shape"][0])
    n = (T - 1) * 80 + 1                                         # x_length giving T frames at 5 ms
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[n])
    assert b.total_frames == T
    streams = []
    for name in ("mgc", "lf0", "bap"):
        _, dim = (int(v) for v in g[name + "_shape"])
        a = cmp_stream(int(g[name + "_seed"]), T, dim, bool(g[name + "_holes"]))
        streams.append((torch.from_numpy(a).cuda(), CMP_WINDOWS))
    out = b.compose_cmp(streams).cpu().numpy()
    want = np.concatenate([g[k + "_windowed"] for k in ("mgc", "lf0", "bap")], axis=1)
    np.testing.assert_array_equal(out.view(np.uint32), want.view(np.uint32))
    sr, shift, byte, kind = (int(v) for v in g["htk_args"])
    assert W.htk_header(T, sr, shift, byte, kind) == bytes(g["htk_header"])
    b.close()
    # ragged batch: clamping happens at each utterance's own first / last frame
    lens = [(29 - 1) * 80 + 1, (41 - 1) * 80 + 1]
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=lens)
    parts = [cmp_stream(7, 29, 4, True), cmp_stream(8, 41, 4, True)]
    got = b.compose_cmp([(torch.from_numpy(np.concatenate(parts)).cuda(), CMP_WINDOWS)]).cpu().numpy()
    ref = np.concatenate([oracle.window_stream(p, CMP_WINDOWS) for p in parts])
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fs,ap_dim", [(16000, 25), (48000, 25), (16000, 24), (48000, 9), (16000, 41)])
def test_recipe_decode_against_oracle(gpu, oracle, fs, ap_dim):
    """The synth CLI's way back from float32 lf0 / mgc / bap (synth.cpp:151-256):
    exp(lf0), DecodeSpectralEnvelope with the c0 offset and 1e-4 scale, mgc2sp of the CLI's SPTK port for bap."""
    torch, W, ctx = gpu
    from test_golden import recipe_pack
    xs = [sd.make_utterance(i, fs, duration=d) for i, d in ((31, 0.6), (32, 0.9))]
    rs = [oracle_chain(oracle, x, fs) for x in xs]
    F = rs[0]["F"]
    sp, ap, f0 = cat(rs, "sp"), cat(rs, "ap"), cat(rs, "f0")
    lf0, mgc, bap = recipe_pack(oracle, f0, sp, ap, fs, F, 50, ap_dim)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    g_f0, g_sp, g_ap = (v.cpu().numpy() for v in b.recipe_decode(dev(lf0), dev(mgc), dev(bap)))
    o_f0, o_sp, o_ap = oracle.recipe_decode(lf0, mgc, bap, fs, F)
    order = ap_dim - 1 if ap_dim % 2 else ap_dim
    assert ((g_f0 > 0) == (o_f0 > 0)).all()
    np.testing.assert_allclose(g_f0, o_f0, rtol=1e-14, atol=0)
    np.testing.assert_allclose(g_sp, o_sp, rtol=1e-10, atol=0)
    np.testing.assert_allclose(g_ap[:, :order], o_ap[:, :order], rtol=1e-11, atol=0)
    assert (g_ap[:, order:] == 0).all() and (o_ap[:, order:] == 0).all()
    # the decoded set drives Synthesis like any other (bins beyond the order are clamped to 0.001 there)
    y = b.synthesize(dev(g_f0), dev(g_sp), dev(g_ap)).cpu().numpy()
    yo = np.concatenate([oracle.synthesis(o_f0[a:e], o_sp[a:e], o_ap[a:e], F, 5.0, fs)
                         for a, e in zip(b.frame_offsets[:-1], b.frame_offsets[1:])])
    np.testing.assert_allclose(y, yo, atol=1e-8, rtol=0)
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["zeros", "tiny", "clipped", "long40s"])
def test_unusual_inputs(gpu, oracle, name):
    """Digital silence, a 1e-6 amplitude, hard clipping, and a 40 s utterance (DIO's reference FFT size 2^20,
    8001 frames): whole chain against the oracle, all outputs finite."""
    torch, W, ctx = gpu
    fs = 16000
    x = {"zeros": lambda: np.zeros(8000), "tiny": lambda: sd.make_utterance(3, fs, duration=0.5) * 1e-6,
         "clipped": lambda: np.clip(sd.make_utterance(6, fs, duration=1.0) * 10, -1, 1),
         "long40s": lambda: sd.make_utterance(5, fs, duration=40.0)}[name]()
    r = oracle_chain(oracle, x, fs)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    t, f0, sp, ap = b.analyze(torch.from_numpy(np.ascontiguousarray(x)).cuda())
    y = b.synthesize(f0, sp, ap)
    f0, sp, ap, y = (v.cpu().numpy() for v in (f0, sp, ap, y))
    assert all(np.isfinite(v).all() for v in (f0, sp, ap, y))
    assert ((f0 > 0) == (r["f0"] > 0)).all()
    np.testing.assert_allclose(f0, r["f0"], atol=F0_TOL, rtol=0)
    sp_close(sp, r["sp"])
    np.testing.assert_allclose(ap, r["ap"], atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(y, r["y"], atol=Y_TOL, rtol=0)
    b.close()


@pytest.mark.gpu
def test_errors_are_reported_not_swallowed(gpu):
    """Unsupported sizes and bad arguments come back as error codes (raised by the Python mirror), never as a
    silent fallback or a wrong answer."""
    torch, W, ctx = gpu
    fs = 16000
    n = 8000
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    # CheapTrick / Synthesis sizes outside {512, 1024, 2048, 4096}
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0, fft_size=256), x_lengths=[n])
    t, f0 = b.dio(x)
    with pytest.raises(RuntimeError):
        b.cheaptrick(x, t, f0)
    b.close()
    with pytest.raises(RuntimeError, match="no samples"):  # as the reference's CLI refuses an empty wav (analysis.cpp:252-259)
        W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[n, 0, n])
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[n])
    T = b.total_frames
    lf0 = torch.zeros(T, dtype=torch.float32, device="cuda")
    mgc = torch.zeros(T, 50, dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError):                      # mgc2sp order beyond one wavefront
        b.recipe_decode(lf0, mgc, torch.zeros(T, 70, dtype=torch.float32, device="cuda"))
    with pytest.raises(RuntimeError):                      # more coefficients than mel points
        b.code_spectral_envelope(torch.ones(T, b.bins, dtype=torch.float64, device="cuda"), 600)
    with pytest.raises(AssertionError):                    # host-side type check of the mirror
        b.analyze(x.float())
    t, f0 = b.dio(x)
    with pytest.raises(RuntimeError):                      # StoneMask clears its output first: in place is refused
        import ctypes as C
        lib = W.load_library()
        rc = lib.WorldMi355StoneMask(b.handle, C.c_void_p(x.data_ptr()), C.c_void_p(t.data_ptr()),
                                     C.c_void_p(f0.data_ptr()), C.c_void_p(f0.data_ptr()))
        if rc != 0:
            raise RuntimeError(lib.WorldMi355LastError())
    b.close()


@pytest.mark.gpu
def test_drop_in_failure_goes_to_the_handler(gpu, pkg, oracle):
    """WorldMi355SetErrorHandler: a drop-in call that fails (CheapTrick at an fft_size the kernels do not have) calls
    the handler instead of abort(), returns, and the next call works as if nothing had happened."""
    import ctypes as C
    capi = pkg.capi
    L = capi._lib()
    seen = []
    H = C.CFUNCTYPE(None, C.c_char_p, C.c_int, C.c_char_p, C.c_void_p)
    cb = H(lambda where, code, msg, user: seen.append((where.decode(), code)))
    L.WorldMi355SetErrorHandler(cb, None)
    try:
        x = sd.make_utterance(90, 16000, duration=0.4)
        t, f0 = capi.dio(x, 16000)
        capi.cheaptrick(x, 16000, t, f0, fft_size=256)
        assert seen == [("CheapTrick", 3)]                              # WM_ERR_UNSUPPORTED_FFT
        sp = capi.cheaptrick(x, 16000, t, f0)
        sp_close(sp, oracle.cheaptrick(x, 16000, t, f0))
        assert len(seen) == 1
    finally:
        L.WorldMi355SetErrorHandler(None, None)


@pytest.mark.gpu
def test_synthesis_in_response_chunks(gpu, monkeypatch):
    """The per-pulse responses go through a scratch buffer of bounded size; a batch whose pulses do not fit is
    rendered chunk by chunk.  Same bits whatever the chunk size."""
    torch, W, ctx = gpu
    fs = 16000
    xs = [sd.make_utterance(i, fs, duration=d) for i, d in ((61, 1.3), (62, 0.7), (63, 2.1))]
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x) for x in xs])
    t, f0, sp, ap = b.analyze(torch.from_numpy(np.concatenate(xs)).cuda())
    y_ref = b.synthesize(f0, sp, ap).clone()
    for mb in ("1", "3"):                                  # 128 and 384 pulses per chunk
        monkeypatch.setenv("WORLD_MI355_SCRATCH_MB", mb)
        assert torch.equal(b.synthesize(f0, sp, ap), y_ref)
        assert torch.equal(b.analyze_synthesize(torch.from_numpy(np.concatenate(xs)).cuda())[4], y_ref)
    b.close()


def test_synthesis_split_with_a_short_pulse_list(gpu):
    """Synthesis alone splits a batch of 16 or more utterances and 4 Mi output samples into two parts.  With few
    pulses (low-pitched, fully voiced: fewer than 32 Ki in all) each part is one piece of the response scratch, and
    the second part lands in the scratch's second half: the layout must hold both (it held one; the second part's
    responses went past the end).  Every utterance must come out as from a batch of its own, bit for bit."""
    torch, W, ctx = gpu
    fs, fp, n_utt = 48000, 5.0, 18
    p = W.default_params(fs, fp)
    rng = np.random.default_rng(5)
    T = [int(rng.integers(1000, 1300)) for _ in range(n_utt)]
    Y = [int((t - 1) * fp / 1000.0 * fs) + 1 for t in T]
    assert sum(Y) >= 4 << 20
    bs = W.WorldBatch(ctx, p, f0_lengths=T, y_lengths=Y)
    H = bs.fft_size // 2 + 1
    g = torch.Generator(device="cuda").manual_seed(11)
    nt = sum(T)
    f0 = 70.0 + 25.0 * torch.rand(nt, dtype=torch.float64, device="cuda", generator=g)
    k = torch.arange(H, dtype=torch.float64, device="cuda")
    sp = (1e-3 * torch.exp(-k / 300.0))[None, :] * (0.5 + torch.rand(nt, H, dtype=torch.float64, device="cuda", generator=g))
    ap = 0.05 + 0.9 * torch.rand(nt, H, dtype=torch.float64, device="cuda", generator=g) * (k / H)[None, :]
    y = bs.synthesize(f0, sp, ap).clone()
    assert bool(torch.isfinite(y).all()) and float(y.abs().max()) > 0
    fo, yo = np.concatenate([[0], np.cumsum(T)]), np.concatenate([[0], np.cumsum(Y)])
    for u in range(n_utt):
        b1 = W.WorldBatch(ctx, p, f0_lengths=[T[u]], y_lengths=[Y[u]])
        fr = slice(int(fo[u]), int(fo[u + 1]))
        y1 = b1.synthesize(f0[fr].contiguous(), sp[fr].contiguous(), ap[fr].contiguous())
        assert torch.equal(y1, y[int(yo[u]):int(yo[u + 1])]), u
        b1.close()
    bs.close()


def test_pcm16_conversions(gpu):
    """wavread's s / 32768 and wavwrite's clamp(int(y * 32767)) (test/audioio.cpp:236-249, :160-167) over a batch."""
    torch, W, ctx = gpu
    lens = [1000, 37, 4096 + 3]
    b = W.WorldBatch(ctx, W.default_params(16000, 5.0), x_lengths=lens)
    rng = np.random.default_rng(5)
    pcm = rng.integers(-32768, 32768, size=sum(lens), dtype=np.int16)
    pcm[:4] = [-32768, 32767, 0, -1]
    x = b.samples_from_pcm16(torch.from_numpy(pcm).cuda()).cpu().numpy()
    np.testing.assert_array_equal(x, pcm.astype(np.float64) / 32768.0)
    y = rng.uniform(-1.3, 1.3, size=int(b.total_out))
    y[:8] = [0.0, -0.0, 1.0, -1.0, 32767.5 / 32767.0, -32768.9 / 32767.0, 0.99999, np.nan]
    got = b.samples_to_pcm16(torch.from_numpy(y).cuda()).cpu().numpy()
    with np.errstate(invalid="ignore"):
        want = np.clip(np.trunc(y * 32767.0), -32768, 32767)
    want[np.isnan(want)] = 0
    np.testing.assert_array_equal(got, want.astype(np.int16))
    b.close()


def test_host_pipeline_gives_the_device_results(gpu, pkg):
    """hts-train-world_amd/pipeline.py: int16 in, float32 features and int16 resynthesis out, three streams, two
    slots -- the same numbers as the resident API converted afterwards, whatever the overlap, step after step."""
    torch, W, ctx = gpu
    pl, recipe = pkg.pipeline, pkg.recipe
    fs = 16000
    sets = [[sd.make_utterance(120 + 3 * k + j, fs, duration=d) for j, d in enumerate((0.5, 0.9, 0.7))] for k in range(4)]
    lens = [len(x) for x in sets[0]]
    pipe = pl.HostPipeline(ctx, W.default_params(fs, 5.0), lens, synthesis=True)
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=lens)
    want = []
    for xs in sets:
        t, f0, sp, ap, y = b.analyze_synthesize(torch.from_numpy(np.concatenate(xs)).cuda())
        y16 = np.clip(np.trunc(y.cpu().numpy() * 32767.0), -32768, 32767).astype(np.int16)
        want.append((f0.float().cpu().numpy(), sp.float().cpu().numpy(), ap.float().cpu().numpy(), y16))
    got, prev = [], None
    pipe.input_buffer()[:] = pl.to_int16(np.concatenate(sets[0]))
    pipe.feed()
    for k in range(len(sets)):                        # one upload ahead, two steps in flight
        if k + 1 < len(sets):
            pipe.input_buffer()[:] = pl.to_int16(np.concatenate(sets[k + 1]))
            pipe.feed()
        slot = pipe.submit()
        if prev is not None:
            got.append(tuple(a.copy() for a in pipe.result(prev)))
        prev = slot
    got.append(tuple(a.copy() for a in pipe.result(prev)))
    # and the simple form: submit() feeds by itself when nothing is fed
    pipe.input_buffer()[:] = pl.to_int16(np.concatenate(sets[1]))
    again = pipe.result(pipe.submit())
    for a, c in zip(again, want[1]):
        np.testing.assert_array_equal(a, c)
    for g, w in zip(got, want):
        for a, c in zip(g, w):
            np.testing.assert_array_equal(a, c)
    up, down = pipe.bytes_per_step()
    assert up == 2 * sum(lens) and down > 4 * int(b.total_frames) * 2 * b.bins
    pipe.close()
    b.close()


def test_host_pipeline_coded_features(gpu, pkg):
    """The same pipeline handing down what the recipe's own call writes (`analysis ... 5 F 50 25`): float32 lf0 /
    mgc[50] / bap[25] -- the resident coder's numbers, 300 B per frame on the wire instead of 4104."""
    torch, W, ctx = gpu
    pl = pkg.pipeline
    fs = 16000
    sets = [[sd.make_utterance(140 + 3 * k + j, fs, duration=d) for j, d in enumerate((0.6, 0.8))] for k in range(3)]
    lens = [len(x) for x in sets[0]]
    pipe = pl.HostPipeline(ctx, W.default_params(fs, 5.0), lens, synthesis=True, coded=(50, 25))
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=lens)
    prev, got = None, []
    for xs in sets:
        pipe.input_buffer()[:] = pl.to_int16(np.concatenate(xs))
        slot = pipe.submit()
        if prev is not None:
            got.append(tuple(a.copy() for a in pipe.result(prev)))
        prev = slot
    got.append(tuple(a.copy() for a in pipe.result(prev)))
    for xs, g in zip(sets, got):
        x = torch.from_numpy(pl.to_int16(np.concatenate(xs))).cuda().to(torch.float64) / 32768.0
        t, f0, sp, ap, y = b.analyze_synthesize(x)
        lf0, mgc, bap = b.recipe_features(f0, sp, ap, 50, 25)
        np.testing.assert_array_equal(g[0], lf0.cpu().numpy())
        np.testing.assert_array_equal(g[1], mgc.cpu().numpy())
        np.testing.assert_array_equal(g[2], bap.cpu().numpy())
        assert g[1].shape[1] == 50 and g[2].shape[1] == 25
    up, down = pipe.bytes_per_step()
    assert down == 4 * int(b.total_frames) * 76 + 2 * int(b.total_out)
    pipe.close()
    b.close()


def test_device_memory_is_bounded_over_many_batches(tmp_path):
    """Batches of ever different sizes (what the drop-in API creates, one per call signature) take their device
    memory from the library's cache of freed blocks: with the cache limited to 64 MB, two hundred batches at 16 and
    48 kHz, each analysed, resynthesised and destroyed, must leave the device where it was -- no leak in any stage
    workspace, nothing kept beyond the limit."""
    code = r'''
import importlib, sys
sys.path.insert(0, %r)
import numpy as np, torch
pkg = importlib.import_module("hts-train-world_amd")
W, sd = pkg.world, pkg.synth_data
ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(5)
def cycle(k):
    fs = 48000 if k %% 5 == 0 else 16000
    lens = [int(fs * d) for d in rng.uniform(0.3, 1.2, size=int(rng.integers(1, 4)))]
    x = torch.from_numpy(np.concatenate([sd.make_utterance(300 + k, fs, duration=n / fs)[:n] for n in lens])).cuda()
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=lens)
    t, f0, sp, ap, y = b.analyze_synthesize(x)
    assert torch.isfinite(y).all()
    b.harvest(x)
    b.close()
for k in range(20):
    cycle(k)
torch.cuda.synchronize(); torch.cuda.empty_cache()
free0 = torch.cuda.mem_get_info()[0]
for k in range(20, 220):
    cycle(k)
torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("DELTA_MB", (free0 - free1) / 2**20)
''' % ROOT
    env = dict(os.environ, WORLD_MI355_CACHE_MB="64")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    delta = float([l for l in r.stdout.splitlines() if l.startswith("DELTA_MB")][0].split()[1])
    assert delta < 512.0, "device memory grew by %.0f MB over 200 batches" % delta


def test_fast_log_exp_against_numpy(gpu):
    """wm_log / wm_exp (csrc/fastmath.hpp: the per-bin logarithms and exponentials of CheapTrick, the pulse kernel, D4C's
    rows and the codec) against numpy in float64 over the ranges the kernels feed them and the special cases they pass
    on to the library: within 2 ulp (the host build of the same source measures 0.8 / 0.9 ulp against long double)."""
    import ctypes as C
    torch, W, ctx = gpu
    lib = C.CDLL(os.path.join(os.path.dirname(__file__), "hooks", "libfastmath_hook.so"))
    vp = C.c_void_p
    lib.FastmathHook.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    rng = np.random.default_rng(11)
    pos = np.exp(rng.uniform(-100.0, 40.0, 400_000)) * rng.uniform(0.5, 2.0, 400_000)      # 1e-44 .. 1e17
    near1 = 1.0 + rng.uniform(-1e-3, 1e-3, 100_000)
    expo = np.concatenate([rng.uniform(-90.0, 40.0, 300_000), rng.uniform(-2000.0, 2000.0, 100_000)])
    special = np.array([0.0, -0.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 1e-310, 2.2250738585072014e-308,
                        1.7976931348623157e308, 1.0, 709.0, -745.0, 800.0, -800.0, 700.0, -700.0])
    big = np.array([2.0 ** 51 + 0.5, 2.0 ** 52 + 1.0, 2.0 ** 53, -2.0 ** 53, 1e300, -1e300, 2.0 ** 40 + 0.25])
    x = np.concatenate([pos, near1, expo, special, big])
    xd = torch.from_numpy(x).cuda()
    lg, ex, sn, cs, sq = (torch.empty_like(xd) for _ in range(5))
    torch.cuda.synchronize()
    rc = lib.FastmathHook(vp(torch.cuda.current_stream().cuda_stream), len(x), vp(xd.data_ptr()), vp(lg.data_ptr()),
                          vp(ex.data_ptr()), vp(sn.data_ptr()), vp(cs.data_ptr()), vp(sq.data_ptr()))
    assert rc == 0
    lg, ex, sn, cs, sq = (v.cpu().numpy() for v in (lg, ex, sn, cs, sq))
    with np.errstate(all="ignore"):
        want_l, want_e = np.log(x), np.exp(x)

    def ulps(got, want):
        ok = np.isfinite(want) & (want != 0)
        return np.abs(got[ok] - want[ok]) / np.spacing(np.abs(want[ok]))
    n1 = len(pos) + len(near1)
    assert ulps(lg[:n1], want_l[:n1]).max() <= 2.0
    sel = slice(n1, n1 + 300_000)
    assert ulps(ex[sel], want_e[sel]).max() <= 2.0
    # wm_sincospi on (-2000, 2000) half-turns: the reduction x - 2 round(x / 2) is exact in float64, the rest in
    # extended precision on the host
    sel = slice(n1, n1 + len(expo))
    xr = (x[sel] - 2.0 * np.round(x[sel] / 2.0)).astype(np.longdouble)
    ws, wc = np.sin(np.pi * xr), np.cos(np.pi * xr)            # np.pi is a double: its 1.2e-16 enters as 4e-17 * |pi x|
    assert np.abs(sn[sel] - ws.astype(np.float64)).max() < 6e-16 and np.abs(cs[sel] - wc.astype(np.float64)).max() < 6e-16
    # special cases: the library's answers (the same NaNs and infinities, zero for underflow)
    tail = slice(n1 + len(expo), n1 + len(expo) + len(special))
    # wm_sincospi without a branch: even integers from 2^53 on, the quadrant taken in floating point below that
    sb, cb = sn[-len(big):], cs[-len(big):]
    np.testing.assert_allclose(np.abs(sb), [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, np.sqrt(0.5)], atol=2e-16)
    np.testing.assert_allclose(cb, [0.0, -1.0, 1.0, 1.0, 1.0, 1.0, np.sqrt(0.5)], atol=2e-16)
    assert sb[0] == 1.0 and sb[6] > 0
    assert np.isnan(sn[tail][3]) and np.isnan(cs[tail][4]) and np.isnan(sn[tail][5])       # +-inf, NaN
    # wm_sqrt (the pulse kernel's sqrt(1 - cos^2)): within 1 ulp on normal arguments, 0 for zero and below
    assert ulps(sq[:n1], np.sqrt(x[:n1])).max() <= 1.0
    neg = x < 0
    assert np.all(sq[neg] == 0.0) and sq[n1 + len(expo)] == 0.0
    np.testing.assert_array_equal(np.isnan(lg[tail]), np.isnan(want_l[tail]))
    np.testing.assert_array_equal(np.isnan(ex[tail]), np.isnan(want_e[tail]))
    fin = ~np.isnan(want_l[tail])
    np.testing.assert_allclose(lg[tail][fin], want_l[tail][fin], rtol=4e-16)
    fin = ~np.isnan(want_e[tail])
    np.testing.assert_allclose(ex[tail][fin], want_e[tail][fin], rtol=4e-16)


@pytest.mark.parametrize("m", [16, 32])
def test_peel_largest_against_numpy(gpu, m):
    """csrc/peel.hpp as d4c_kernel uses it (tests/hooks/libpeel_hook.so): the sum of all but the K largest of
    64 x (m + 1) values, on noise, on a main lobe of neighbouring bins, on ties (all equal, blocks of equal values,
    zeros) and with fewer values than K -- compared with a full sort; the per-lane counts add up to K."""
    import ctypes as C
    torch, W, ctx = gpu
    lib = C.CDLL(os.path.join(os.path.dirname(__file__), "hooks", "libpeel_hook.so"))
    lib.PeelHook.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(11)
    n = 64 * m + 1                                   # values of a case: bins 0 .. fft/2
    cases = []
    for _ in range(40):
        cases.append(rng.exponential(size=n))
    for _ in range(20):
        cases.append(100.0 * np.exp(-np.arange(n) / rng.uniform(3, 30)) + rng.exponential(size=n))
    for _ in range(10):
        cases.append(rng.exponential(size=n) + 50.0 * (np.arange(n) % rng.integers(40, 200) < 3))
    cases.append(np.full(n, 0.25))                   # all equal
    cases.append(np.zeros(n))
    cases.append(np.repeat(rng.exponential(size=(n + 63) // 64), 64)[:n])        # runs of 64 equal values
    cases.append(np.round(rng.exponential(size=n) * 4) / 4)                      # few distinct values
    cases.append(np.concatenate([np.full(70, 7.0), np.full(n - 70, 1.0)]))       # more ties at the top than K
    vals = np.full((len(cases), 64, m + 1), -1.0)
    for c, v in enumerate(cases):                    # lane l holds bins l + 64 k (k < m); bin 64 m on lane 0
        vals[c, :, :m] = v[:64 * m].reshape(m, 64).T
        vals[c, 0, m] = v[64 * m]
    d_vals = torch.from_numpy(vals).cuda()
    for K in (65, 1, 64, 200):
        low = torch.empty(len(cases), dtype=torch.float64, device="cuda")
        taken = torch.empty(len(cases), 64, dtype=torch.int32, device="cuda")
        rc = lib.PeelHook(C.c_void_p(torch.cuda.current_stream().cuda_stream), m, len(cases), K,
                          C.c_void_p(d_vals.data_ptr()), C.c_void_p(low.data_ptr()), C.c_void_p(taken.data_ptr()))
        assert rc == 0
        torch.cuda.synchronize()
        want = np.array([np.sort(v)[:n - K].sum() for v in cases])
        np.testing.assert_allclose(low.cpu().numpy(), want, rtol=1e-12, atol=1e-300)
        assert np.all(taken.cpu().numpy().sum(axis=1) == K)


def test_block_events_list_and_direct_paths(gpu):
    """csrc/fftconv.hpp conv_block_events (tests/hooks/libevents_hook.so): the four zero-crossing passes of a filtered
    block (dio.cpp:357-435) against numpy, through the ordered-list path, through the direct path that blocks with
    more events than the lists hold take (forced with a small capacity, and reached for real by a signal that
    alternates around zero every sample), bit for bit the same."""
    import ctypes as C
    torch, W, ctx = gpu
    lib = C.CDLL(os.path.join(os.path.dirname(__file__), "hooks", "libevents_hook.so"))
    lib.EventsHook.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    slot, cap = lib.EventsHookSlot(), lib.EventsHookListCap()
    rng = np.random.default_rng(3)
    step, blocks = 1729, 6
    ylen = step * blocks - 5
    y = np.zeros(step * blocks + 2)
    n = np.arange(len(y))
    sigs = {"tones": np.sin(2 * np.pi * 0.013 * n) + 0.4 * np.sin(2 * np.pi * 0.071 * n + 1.0) + 0.01 * rng.standard_normal(len(y)),
            "alternating": np.where(n % 2 == 0, 1.0, -1.0) * (1.0 + 0.3 * rng.random(len(y))),
            "noise": rng.standard_normal(len(y))}

    def numpy_events(y):
        out = []
        for ty in range(4):
            v = y if ty == 0 else (-y if ty == 1 else (np.diff(y) if ty == 2 else -np.diff(y)))
            lim = ylen - 1 if ty < 2 else ylen - 2
            i = np.nonzero((v[:-1] > 0) & (v[1:] <= 0))[0]
            i = i[i < lim]
            out.append((i + 1) - v[i] / (v[i + 1] - v[i]))
        return out

    def run(y, list_cap):
        s = np.stack([y[b * step:b * step + step + 2] for b in range(blocks)])
        d_s = torch.from_numpy(np.ascontiguousarray(s)).cuda()
        cnt = torch.zeros(blocks, 4, dtype=torch.int32, device="cuda")
        slots = torch.zeros(4, blocks * slot, dtype=torch.float64, device="cuda")
        rc = lib.EventsHook(C.c_void_p(torch.cuda.current_stream().cuda_stream), blocks, step, ylen, list_cap,
                            C.c_void_p(d_s.data_ptr()), C.c_void_p(cnt.data_ptr()), C.c_void_p(slots.data_ptr()))
        assert rc == 0
        torch.cuda.synchronize()
        cnt, slots = cnt.cpu().numpy(), slots.cpu().numpy()
        return [np.concatenate([slots[ty, b * slot:b * slot + cnt[b, ty]] for b in range(blocks)]) for ty in range(4)], cnt

    for name, y in sigs.items():
        want = numpy_events(y)
        got, cnt = run(y, cap)
        forced, _ = run(y, 8)
        if name == "alternating":
            assert cnt.max() > cap                       # the real overflow: the direct path was taken
        for ty in range(4):
            np.testing.assert_array_equal(got[ty], want[ty], err_msg="%s kind %d" % (name, ty))
            np.testing.assert_array_equal(forced[ty], want[ty], err_msg="%s kind %d (direct path)" % (name, ty))
