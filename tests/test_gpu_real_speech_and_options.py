"""The HIP path on the reference's OWN inputs and away from the default options, against vectors of the compiled
reference (tests/golden/real_*.npz, options_*.npz; written by oracle/gen_golden.py from
externs/WORLD_v2/wav_test/arctic_a0001.wav -- SURVEY.md's config 1 -- and test/vaiueo2d.wav) and the oracle.

Real speech is where the thresholded branches of Dio (dio.cpp:145-147, 206, 459-463), StoneMask
(stonemask.cpp:128, 186, 202) and Harvest (harvest.cpp:250-252, 371, 610-611, 664) are taken: creak, onsets,
near-silence.  Each case goes through the batched device API, the drop-in C ABI (host pointers, `double**`
rows) and, in tests/test_cli_relink.py, the reference's own CLIs relinked against the library.

Tolerances are those of tests/test_gpu_parity.py: |dF0| < 1e-6 Hz with identical V/UV, sp relative 1e-6,
ap / y absolute 1e-8 (bars of BASELINE.json: 0.1 Hz, RMSE 1e-5).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

F0_TOL, AP_TOL, Y_TOL = 1e-6, 1e-8, 1e-8


def sp_close(a, b):
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-13)


def checks(a):
    return np.array([a.sum(), (a * a).sum(), np.abs(a).max()])


def same_voicing(a, b):
    assert ((a > 0) == (b > 0)).all(), "V/UV decisions differ at frames %s" % np.nonzero((a > 0) != (b > 0))[0][:10]


REAL = ["real_arctic_a0001", "real_vaiueo2d"]


@pytest.mark.parametrize("name", REAL)
def test_real_speech_batched_api(gpu, name):
    """Dio -> StoneMask -> CheapTrick -> D4C (threshold 0 and 0.85) -> Synthesis, and Harvest, on a batch of one."""
    torch, W, ctx = gpu
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    x = g["x_i16"].astype(np.float64) / 32768.0
    fs, fp, F = int(g["fs"]), float(g["frame_period"]), int(g["fft_size"])
    fs_, ss = int(g["frame_step"]), int(g["sample_step"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    xc = dev(x)
    b = W.WorldBatch(ctx, W.default_params(fs, fp), x_lengths=[len(x)])
    assert b.fft_size == F
    # stage by stage, each fed the reference's own input (a difference cannot hide behind an earlier one)
    t, f0_dio = b.dio(xc)
    np.testing.assert_array_equal(t.cpu().numpy(), g["t"])
    same_voicing(f0_dio.cpu().numpy(), g["f0_dio"])
    np.testing.assert_allclose(f0_dio.cpu().numpy(), g["f0_dio"], atol=F0_TOL, rtol=0)
    f0 = b.stonemask(xc, dev(g["t"]), dev(g["f0_dio"]))
    np.testing.assert_allclose(f0.cpu().numpy(), g["f0"], atol=F0_TOL, rtol=0)
    th, f0_hv = b.harvest(xc)
    np.testing.assert_array_equal(th.cpu().numpy(), g["t"])
    same_voicing(f0_hv.cpu().numpy(), g["f0_harvest"])
    np.testing.assert_allclose(f0_hv.cpu().numpy(), g["f0_harvest"], atol=F0_TOL, rtol=0)
    f0_hs = b.stonemask(xc, dev(g["t"]), dev(g["f0_harvest"]))
    np.testing.assert_allclose(f0_hs.cpu().numpy(), g["f0_harvest_sm"], atol=F0_TOL, rtol=0)
    # the chain as the analysis CLI runs it (threshold 0), end to end on its own intermediate results
    t2, f02, sp, ap = b.analyze(xc)
    y = b.synthesize(f02, sp, ap)
    f02, sp, ap, y = (v.cpu().numpy() for v in (f02, sp, ap, y))
    same_voicing(f02, g["f0"])
    np.testing.assert_allclose(f02, g["f0"], atol=F0_TOL, rtol=0)
    sp_close(sp[::fs_], g["sp_sub"])
    np.testing.assert_allclose(checks(sp), g["sp_check"], rtol=1e-7)
    np.testing.assert_allclose(ap[::fs_], g["ap_sub"], atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(checks(ap), g["ap_check"], rtol=1e-7)
    np.testing.assert_allclose(y[::ss], g["y_sub"], atol=Y_TOL, rtol=0)
    np.testing.assert_allclose(checks(y), g["y_check"], rtol=1e-6)
    b.close()
    # the library's default threshold (LoveTrain decides) and Harvest's contour through the same back end
    b = W.WorldBatch(ctx, W.default_params(fs, fp, d4c_threshold=0.85), x_lengths=[len(x)])
    ap85 = b.d4c(xc, dev(g["t"]), dev(g["f0"])).cpu().numpy()
    np.testing.assert_allclose(ap85[::fs_], g["ap85_sub"], atol=AP_TOL, rtol=0)
    np.testing.assert_allclose(checks(ap85), g["ap85_check"], rtol=1e-7)
    fh = dev(g["f0_harvest_sm"])
    sp_h = b.cheaptrick(xc, dev(g["t"]), fh)
    ap_h = b.d4c(xc, dev(g["t"]), fh)
    y_h = b.synthesize(fh, sp_h, ap_h).cpu().numpy()
    np.testing.assert_allclose(checks(sp_h.cpu().numpy()), g["sp_h_check"], rtol=1e-7)
    np.testing.assert_allclose(checks(ap_h.cpu().numpy()), g["ap_h_check"], rtol=1e-7)
    np.testing.assert_allclose(y_h[::ss], g["y_h_sub"], atol=Y_TOL, rtol=0)
    b.close()


@pytest.mark.parametrize("name", REAL)
def test_real_speech_drop_in_c_abi(pkg, name):
    """The same through WORLD's own entry points: host pointers, one utterance per call, `double**` rows."""
    C = pkg.capi
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    x = g["x_i16"].astype(np.float64) / 32768.0
    fs, fp, F = int(g["fs"]), float(g["frame_period"]), int(g["fft_size"])
    fs_, ss = int(g["frame_step"]), int(g["sample_step"])
    t, f0_dio = C.dio(x, fs, fp)
    np.testing.assert_array_equal(t, g["t"])
    same_voicing(f0_dio, g["f0_dio"])
    np.testing.assert_allclose(f0_dio, g["f0_dio"], atol=F0_TOL, rtol=0)
    f0 = C.stonemask(x, fs, t, f0_dio)
    np.testing.assert_allclose(f0, g["f0"], atol=F0_TOL, rtol=0)
    th, f0_hv = C.harvest(x, fs, fp)
    same_voicing(f0_hv, g["f0_harvest"])
    np.testing.assert_allclose(f0_hv, g["f0_harvest"], atol=F0_TOL, rtol=0)
    assert C.cheaptrick_fft_size(fs) == F
    sp = C.cheaptrick(x, fs, t, f0)
    sp_close(sp[::fs_], g["sp_sub"])
    ap = C.d4c(x, fs, t, f0, F, 0.0)
    np.testing.assert_allclose(ap[::fs_], g["ap_sub"], atol=AP_TOL, rtol=0)
    ap85 = C.d4c(x, fs, t, f0, F)
    np.testing.assert_allclose(ap85[::fs_], g["ap85_sub"], atol=AP_TOL, rtol=0)
    y = C.synthesis(f0, sp, ap, F, fp, fs)
    np.testing.assert_allclose(y[::ss], g["y_sub"], atol=Y_TOL, rtol=0)
    np.testing.assert_allclose(checks(y), g["y_check"], rtol=1e-6)


def test_real_speech_inside_a_batch(gpu):
    """The real utterance between synthetic ones and a cut of itself: the batch's results for it are the
    single-utterance results (utterances of a batch never exchange data)."""
    import importlib
    torch, W, ctx = gpu
    sd = importlib.import_module("hts-train-world_amd.synth_data")
    g = np.load(os.path.join(GOLDEN, "real_arctic_a0001.npz"))
    x = g["x_i16"].astype(np.float64) / 32768.0
    fs = int(g["fs"])
    xs = [sd.make_utterance(61, fs, duration=1.3), x, sd.make_utterance(62, fs, duration=0.9), x[5000:30000]]
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(v) for v in xs])
    t, f0, sp, ap = b.analyze(torch.from_numpy(np.concatenate(xs)).cuda())
    y = b.synthesize(f0, sp, ap)
    f0u, spu, apu, yu = b.split_frames(f0)[1], b.split_frames(sp)[1], b.split_frames(ap)[1], b.split_out(y)[1]
    b1 = W.WorldBatch(ctx, W.default_params(fs, 5.0), x_lengths=[len(x)])
    t1, f01, sp1, ap1 = b1.analyze(torch.from_numpy(x).cuda())
    y1 = b1.synthesize(f01, sp1, ap1)
    np.testing.assert_allclose(f0u.cpu().numpy(), f01.cpu().numpy(), atol=1e-9, rtol=0)
    same_voicing(f0u.cpu().numpy(), f01.cpu().numpy())
    np.testing.assert_allclose(spu.cpu().numpy(), sp1.cpu().numpy(), rtol=1e-9, atol=1e-16)
    np.testing.assert_allclose(apu.cpu().numpy(), ap1.cpu().numpy(), atol=1e-10, rtol=0)
    np.testing.assert_allclose(yu.cpu().numpy(), y1.cpu().numpy(), atol=1e-10, rtol=0)
    np.testing.assert_allclose(f0u.cpu().numpy(), g["f0"], atol=F0_TOL, rtol=0)
    b.close()
    b1.close()


OPTIONS = ["options_16k", "options_22k", "options_48k"]


@pytest.mark.parametrize("name", OPTIONS)
def test_cheaptrick_d4c_synthesis_option_sweep(gpu, pkg, oracle, name):
    """CheapTrickOption.q1 in {-0.15, -0.09, 0} x fft_size in {default, 2 x default, default / 2}
    (cheaptrick.h:16-20, cheaptrick.cpp:191-228: the floor follows the size, frames at or below it are analysed at
    the default f0), D4C with that size for its rows (d4c.cpp:337-397) and Synthesis from the set
    (synthesis.cpp:338-397): batched API and drop-in C ABI against the compiled reference's vectors; full arrays
    against the oracle."""
    torch, W, ctx = gpu
    C = pkg.capi
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    x = g["x_i16"].astype(np.float64) / 32768.0
    fs, fp, fs_, ss = int(g["fs"]), float(g["frame_period"]), int(g["frame_step"]), int(g["sample_step"])
    t, f0 = g["t"], g["f0"]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    xc, tc, fc = dev(x), dev(t), dev(f0)
    for F in (int(v) for v in g["fft_sizes"]):
        floor = 3.0 * fs / (F - 3.0)
        assert abs(C._lib().GetF0FloorForCheapTrick(fs, F) - floor) < 1e-12
        ap_o = oracle.d4c(x, fs, t, f0, F, 0.85)
        for qi, q1 in enumerate(float(v) for v in g["q1"]):
            b = W.WorldBatch(ctx, W.default_params(fs, fp, q1=q1, fft_size=F, d4c_threshold=0.85), x_lengths=[len(x)])
            assert b.fft_size == F and b.bins == F // 2 + 1
            sp = b.cheaptrick(xc, tc, fc)
            spn = sp.cpu().numpy()
            sp_close(spn[::fs_], g["sp_%d_q%d_sub" % (F, qi)])
            np.testing.assert_allclose(checks(spn), g["sp_%d_q%d_check" % (F, qi)], rtol=1e-7)
            sp_close(spn, oracle.cheaptrick(x, fs, t, f0, q1, F))
            sp_c = C.cheaptrick(x, fs, t, f0, q1, F)
            np.testing.assert_array_equal(sp_c, spn)                # one kernel behind both doors
            if qi == 1:
                ap = b.d4c(xc, tc, fc)
                apn = ap.cpu().numpy()
                np.testing.assert_allclose(apn[::fs_], g["ap_%d_sub" % F], atol=AP_TOL, rtol=0)
                np.testing.assert_allclose(checks(apn), g["ap_%d_check" % F], rtol=1e-7)
                np.testing.assert_allclose(apn, ap_o, atol=AP_TOL, rtol=0)
                np.testing.assert_array_equal(C.d4c(x, fs, t, f0, F), apn)
                y = b.synthesize(fc, sp, ap).cpu().numpy()
                np.testing.assert_allclose(y[::ss], g["y_%d_sub" % F], atol=Y_TOL, rtol=0)
                np.testing.assert_allclose(checks(y), g["y_%d_check" % F], rtol=1e-6)
                np.testing.assert_array_equal(C.synthesis(f0, spn, apn, F, fp, fs), y)
                # the whole chain with the option set (Dio + StoneMask in front): f0 does not depend on them
                t2, f02, sp2, ap2 = b.analyze(xc)
                np.testing.assert_allclose(f02.cpu().numpy(), f0, atol=F0_TOL, rtol=0)
                sp_close(sp2.cpu().numpy()[::fs_], g["sp_%d_q%d_sub" % (F, qi)])
            b.close()


def test_option_sweep_in_a_mixed_batch(gpu, oracle):
    """Non-default q1 / fft_size over a batch of several utterances (frames of all of them dealt to the same
    waves): 16 kHz at fft 2048 and 512, against the oracle."""
    import importlib
    torch, W, ctx = gpu
    sd = importlib.import_module("hts-train-world_amd.synth_data")
    fs = 16000
    xs = [sd.make_utterance(i, fs, duration=d) for i, d in ((45, 0.9), (24, 1.4), (66, 0.6))]   # low voices: below 94 Hz
    for F, q1 in ((2048, -0.09), (512, 0.0)):
        b = W.WorldBatch(ctx, W.default_params(fs, 5.0, q1=q1, fft_size=F, d4c_threshold=0.85),
                         x_lengths=[len(v) for v in xs])
        t, f0, sp, ap = b.analyze(torch.from_numpy(np.concatenate(xs)).cuda())
        y = b.synthesize(f0, sp, ap)
        below = 0
        for u, x in enumerate(xs):
            to, f0o = oracle.dio(x, fs)
            f0o = oracle.stonemask(x, fs, to, f0o)
            below += int(((f0o > 0) & (f0o <= 3.0 * fs / (F - 3.0))).sum())
            np.testing.assert_allclose(b.split_frames(f0)[u].cpu().numpy(), f0o, atol=F0_TOL, rtol=0)
            spo = oracle.cheaptrick(x, fs, to, f0o, q1, F)
            apo = oracle.d4c(x, fs, to, f0o, F, 0.85)
            sp_close(b.split_frames(sp)[u].cpu().numpy(), spo)
            np.testing.assert_allclose(b.split_frames(ap)[u].cpu().numpy(), apo, atol=AP_TOL, rtol=0)
            yo = oracle.synthesis(f0o, spo, apo, F, 5.0, fs)
            np.testing.assert_allclose(b.split_out(y)[u].cpu().numpy(), yo, atol=Y_TOL, rtol=0)
        if F == 512:
            assert below > 0                                               # the default-f0 branch is taken
        b.close()


@pytest.mark.parametrize("name", REAL)
def test_real_speech_front_end_options(gpu, oracle, name):
    """The F0 front ends away from their defaults on the reference's own inputs, against the oracle (pinned to the
    compiled reference on the same grid by tests/test_oracle_vs_ref.py): Dio with decimation (speed 2, 4: the
    thresholded contour logic behind a decimated signal), 1 and 10 ms hops, a narrower search range, more channels
    per octave; Harvest with other floors / ceilings and a 1 ms hop; each followed by StoneMask."""
    torch, W, ctx = gpu
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    x = g["x_i16"].astype(np.float64) / 32768.0
    fs = int(g["fs"])
    xc = torch.from_numpy(x).cuda()
    for kw in (dict(speed=2), dict(speed=4), dict(frame_period=1.0), dict(frame_period=10.0),
               dict(f0_floor=50.0, f0_ceil=500.0), dict(channels_in_octave=4.0), dict(allowed_range=0.05)):
        fp = kw.get("frame_period", 5.0)
        p = W.default_params(fs, fp, **{k: v for k, v in kw.items() if k != "frame_period"})
        b = W.WorldBatch(ctx, p, x_lengths=[len(x)])
        t, f0 = b.dio(xc)
        to, fo = oracle.dio(x, fs, **dict(kw, frame_period=fp))
        np.testing.assert_array_equal(t.cpu().numpy(), to)
        same_voicing(f0.cpu().numpy(), fo)
        np.testing.assert_allclose(f0.cpu().numpy(), fo, atol=F0_TOL, rtol=0)
        sm = b.stonemask(xc, t, f0).cpu().numpy()
        np.testing.assert_allclose(sm, oracle.stonemask(x, fs, to, fo), atol=F0_TOL, rtol=0)
        b.close()
    for fp, lo, hi in ((5.0, 50.0, 500.0), (5.0, 100.0, 1000.0), (1.0, 71.0, 800.0)):
        b = W.WorldBatch(ctx, W.default_params(fs, fp, f0_floor=lo, f0_ceil=hi), x_lengths=[len(x)])
        t, f0 = b.harvest(xc)
        to, fo = oracle.harvest(x, fs, fp, lo, hi)
        np.testing.assert_array_equal(t.cpu().numpy(), to)
        same_voicing(f0.cpu().numpy(), fo)
        np.testing.assert_allclose(f0.cpu().numpy(), fo, atol=F0_TOL, rtol=0)
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fs,thr", [(16000, 0.0), (16000, 0.85), (48000, 0.85)])
def test_one_call_forms_with_frames_outside_the_usual_range(gpu, oracle, fs, thr):
    """WorldMi355Analyze / AnalyzeSynthesize run D4C's preparation on a second stream beside CheapTrick and the launch
    for frames outside the usual range (f0 >= fs / 16, windows past half the D4C transform) on a third, from the
    second call on a batch on (the first allocates in the plain order).  With Dio's range opened (40 Hz .. 0.3 fs) a
    batch of a whistle, a growl and ordinary speech has such frames: every call gives the bits of the first, the
    one-call forms those of the stage-by-stage calls, and the rows are the oracle's."""
    import importlib
    torch, W, ctx = gpu
    sd = importlib.import_module("hts-train-world_amd.synth_data")
    rng = np.random.default_rng(17)
    n = int(0.5 * fs)
    tt = np.arange(n) / fs
    whistle = sum(np.sin(2 * np.pi * (fs / 14.0) * h * tt + h) / h for h in (1, 2, 3)) * 0.2   # between fs / 16 and StoneMask's fs / 12
    growl = sum(np.sin(2 * np.pi * 45.0 * h * tt + 0.3 * h) / h for h in range(1, 40)) * 0.1
    xs = [np.round((whistle + 1e-3 * rng.standard_normal(n)) * 32768.0) / 32768.0,
          np.round((growl + 1e-3 * rng.standard_normal(n)) * 32768.0) / 32768.0,
          sd.make_utterance(91, fs, duration=0.6)]
    opt = dict(f0_floor=40.0, f0_ceil=0.3 * fs, d4c_threshold=thr)      # fft 2048 at 16 kHz, 4096 at 48 kHz
    b = W.WorldBatch(ctx, W.default_params(fs, 5.0, **opt), x_lengths=[len(x) for x in xs])
    xc = torch.from_numpy(np.concatenate(xs)).cuda()
    first = [a.clone() for a in b.analyze(xc)]
    f0 = first[1].cpu().numpy()
    fo = np.asarray(b.frame_offsets)
    F = int(b.fft_size)                                    # follows the lowered f0_floor (cheaptrick.cpp:191-194)
    fd = 2 ** (1 + int(np.log2(4.0 * fs / 47.0 + 1.0)))
    rare = (f0 >= fs / 16.0) | ((f0 > 0) & (2 * np.round(2.0 * fs / np.maximum(f0, 47.0)) + 1 > fd // 2) & (fd < 4096))
    assert rare[fo[0]:fo[1]].any(), "the whistle has no frame beyond fs / 16"
    for _ in range(2):                                     # the streamed order from here on
        again = b.analyze(xc)
        for a, c in zip(first, again):
            assert torch.equal(a, c)
    staged_sp = b.cheaptrick(xc, first[0], first[1])
    staged_ap = b.d4c(xc, first[0], first[1])
    assert torch.equal(staged_sp, first[2]) and torch.equal(staged_ap, first[3])
    y = b.synthesize(first[1], first[2], first[3]).clone()
    for _ in range(2):
        t2, f02, sp2, ap2, y2 = b.analyze_synthesize(xc)
        assert torch.equal(f02, first[1]) and torch.equal(sp2, first[2]) and torch.equal(ap2, first[3])
        assert torch.equal(y2, y)
    ap = first[3].cpu().numpy()
    for u, x in enumerate(xs):
        s = slice(fo[u], fo[u + 1])
        want = oracle.d4c(x, fs, first[0].cpu().numpy()[s], f0[s], F, thr)
        np.testing.assert_allclose(ap[s], want, atol=AP_TOL, rtol=0)
    b.close()
