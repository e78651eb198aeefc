"""The reference's own CLIs (externs/WORLD_v2/test/analysis.cpp, synth.cpp; SURVEY.md 8(b) "Callers" and "File
contract") relinked, source unchanged, against libworld_mi355.so, next to the same CLIs linked against the
reference objects.  Both are built by `make -C oracle cli` into oracle/_ref/cli/ (binaries only; they travel
to the GPU box like the other built files).  Each pair runs on the same 16-bit wav and the output files are
compared.  Skipped where the binaries have not been built."""
import importlib
import os
import subprocess
import wave

import numpy as np
import pytest

sd = importlib.import_module("hts-train-world_amd.synth_data")
CLI = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "cli")


def cli(name):
    path = os.path.join(CLI, name)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/cli not built (make -C oracle cli needs /root/reference)")
    return path


def write_wav(path, x, fs):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(fs)
        w.writeframes(np.round(x * 32768.0).astype("<i2").tobytes())


def read_wav(path):
    with wave.open(str(path), "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.int32), w.getframerate()


def run(binary, *args):
    r = subprocess.run([binary, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def analysis_files(binary, wav, out, tag, extra):
    names = [out / f"{tag}.{e}" for e in ("f0", "sp", "ap")]
    run(binary, wav, *names, *extra)
    return [np.fromfile(n, dtype=np.float32) for n in names]


def test_reference_cli_matches_the_recipe_packing_of_the_oracle(oracle, tmp_path):
    """CPU: the file contract (analysis.cpp:292-390) as the tests' recipe_pack() states it."""
    from test_golden import recipe_pack
    fs, F = 16000, 1024
    x = sd.make_utterance(21, fs, duration=1.2)
    wav = tmp_path / "in.wav"
    write_wav(wav, x, fs)
    lf0, mgc, bap = analysis_files(cli("analysis_ref"), wav, tmp_path, "ref", (5, F, 50, 25))
    t, f0 = oracle.dio(x, fs)
    f0 = oracle.stonemask(x, fs, t, f0)
    sp = oracle.cheaptrick(x, fs, t, f0, -0.15, F)
    ap = oracle.d4c(x, fs, t, f0, F, 0.0)
    o_lf0, o_mgc, o_bap = recipe_pack(oracle, f0, sp, ap, fs, F, 50, 25)
    assert lf0.shape == o_lf0.shape and ((lf0 != 0) == (o_lf0 != 0)).all()
    np.testing.assert_allclose(lf0, o_lf0, atol=1e-6, rtol=0)
    np.testing.assert_allclose(mgc.reshape(-1, 50), o_mgc, atol=2e-5, rtol=0)
    np.testing.assert_allclose(bap.reshape(-1, 25), o_bap, atol=2e-5, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("fs,F,dur,index", [(16000, 1024, 2.3, 22), (48000, 2048, 1.1, 23)])
def test_relinked_cli_roundtrip(gpu, tmp_path, fs, F, dur, index):
    """analysis -> files -> synth through the reference's main() on both link lines."""
    x = sd.make_utterance(index, fs, duration=dur)
    wav = tmp_path / "in.wav"
    write_wav(wav, x, fs)
    a_ref, a_gpu = cli("analysis_ref"), cli("analysis_mi355")
    s_ref, s_gpu = cli("synth_ref"), cli("synth_mi355")

    # uncompressed: f0 in Hz, sp / ap rows of F/2+1 float32 (analysis.cpp:360-390)
    rf0, rsp, rap = analysis_files(a_ref, wav, tmp_path, "ref", (5, F))
    gf0, gsp, gap = analysis_files(a_gpu, wav, tmp_path, "gpu", (5, F))
    assert rf0.shape == gf0.shape and rsp.shape == gsp.shape == (len(rf0) * (F // 2 + 1),)
    assert ((rf0 > 0) == (gf0 > 0)).all()
    assert np.abs(rf0.astype(np.float64) - gf0).max() < 0.1            # north_star: max|dF0| < 0.1 Hz
    assert np.sqrt(np.mean((rsp.astype(np.float64) - gsp) ** 2)) < 1e-5
    assert np.sqrt(np.mean((rap.astype(np.float64) - gap) ** 2)) < 1e-5
    # float32 files of doubles that agree to 1e-12: identical but for values next to a float32 rounding boundary (one ulp)
    assert (rf0 != gf0).mean() < 1e-2 and (rsp != gsp).mean() < 1e-2 and (rap != gap).mean() < 1e-2
    np.testing.assert_allclose(gsp, rsp, rtol=3e-7, atol=0)
    np.testing.assert_allclose(gap, rap, rtol=3e-7, atol=1e-12)

    # the recipe's call: coded features (data/Makefile.in:214; analysis.cpp:292-366)
    rl, rm, rb = analysis_files(a_ref, wav, tmp_path, "refc", (5, F, 50, 25))
    gl, gm, gb = analysis_files(a_gpu, wav, tmp_path, "gpuc", (5, F, 50, 25))
    assert rm.shape == gm.shape == (len(rl) * 50,) and rb.shape == gb.shape == (len(rl) * 25,)
    np.testing.assert_allclose(gl, rl, atol=1e-6, rtol=0)
    np.testing.assert_allclose(gm, rm, atol=2e-6, rtol=0)
    np.testing.assert_allclose(gb, rb, atol=2e-6, rtol=0)

    # synth from the reference's uncompressed files on both link lines (synth.cpp:124-262)
    names = [tmp_path / f"ref.{e}" for e in ("f0", "sp", "ap")]
    run(s_ref, *names, tmp_path / "ref_y.wav", 5, F, fs)
    run(s_gpu, *names, tmp_path / "gpu_y.wav", 5, F, fs)
    yr, fr = read_wav(tmp_path / "ref_y.wav")
    yg, fg = read_wav(tmp_path / "gpu_y.wav")
    assert fr == fg == fs and yr.shape == yg.shape
    assert np.abs(yr - yg).max() <= 1 and (yr != yg).mean() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["real_arctic_a0001", "real_vaiueo2d"])
def test_relinked_cli_on_the_reference_wavs(gpu, tmp_path, name):
    """The reference's own two wav payloads (tests/golden/real_*.npz holds their int16 samples) through its own
    main()s on both link lines: uncompressed and coded analysis files, then synth from the reference's files."""
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    fs, F = int(g["fs"]), int(g["fft_size"])
    wav = tmp_path / "in.wav"
    with wave.open(str(wav), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(fs)
        w.writeframes(g["x_i16"].astype("<i2").tobytes())
    a_ref, a_gpu = cli("analysis_ref"), cli("analysis_mi355")
    s_ref, s_gpu = cli("synth_ref"), cli("synth_mi355")
    rf0, rsp, rap = analysis_files(a_ref, wav, tmp_path, "ref", (5, F))
    gf0, gsp, gap = analysis_files(a_gpu, wav, tmp_path, "gpu", (5, F))
    # the CLI's f0 file is what the fixture holds (Dio + StoneMask of the library, float32)
    np.testing.assert_array_equal(rf0, g["f0"].astype(np.float32))
    assert rf0.shape == gf0.shape and rsp.shape == gsp.shape == (len(rf0) * (F // 2 + 1),)
    assert ((rf0 > 0) == (gf0 > 0)).all()
    assert np.abs(rf0.astype(np.float64) - gf0).max() < 1e-3
    # float32 files of doubles that agree to 1e-8 relative (the quiet bins of real speech: the blocked cumulative
    # sum of LinearSmoothing) differ by one float32 ulp where a value sits next to a rounding
    # boundary: never more than that
    assert (rf0 != gf0).mean() < 1e-2 and (rsp != gsp).mean() < 0.15 and (rap != gap).mean() < 1e-2
    # One float32 ulp, plus 2e-14 of the frame's total: LinearSmoothing takes differences of a cumulative sum of the
    # POWER spectrum (common.cpp:38-41, 99-108; harmonic peaks well above the envelope), so a bin 80-110 dB below the
    # frame's strongest carries the rounding of the whole sum -- in the reference (sequential order) as much as here
    # (blocked order); tools/sp_quiet_bins.py: bins more than 90 dB down differ by up to 7e-5 relative, 4e-16
    # absolute, on these two files; the worst case against this bound is 8e-15 of the row sum.
    rows_r, rows_g = rsp.reshape(len(rf0), -1).astype(np.float64), gsp.reshape(len(rf0), -1).astype(np.float64)
    excess = np.abs(rows_g - rows_r) - (3e-7 * rows_r + 2e-14 * rows_r.sum(axis=1, keepdims=True))
    i, j = np.unravel_index(np.argmax(excess), excess.shape)
    assert excess[i, j] <= 0, "frame %d bin %d: %r against %r, row sum %r, f0 %r" % (i, j, rows_g[i, j], rows_r[i, j],
                                                                                   rows_r[i].sum(), rf0[i])
    np.testing.assert_allclose(gap, rap, rtol=3e-7, atol=1e-12)
    # The same in float32 ulps (the distance between the bit patterns of two positive floats), by the bin's level under
    # its frame's strongest -- this is the largest deviation between the two link lines anywhere in the suite, so it
    # gets numbers, not a sentence.
    ulp = np.abs(rows_g.astype(np.float32).view(np.int32).astype(np.int64) - rows_r.astype(np.float32).view(np.int32).astype(np.int64))
    level = rows_r / rows_r.max(axis=1, keepdims=True)
    stats = {"differ": float((ulp > 0).mean()), "more_than_one_ulp": float((ulp > 1).mean())}
    for lo, hi, key in ((1e-3, 2.0, "0..-30dB"), (1e-6, 1e-3, "-30..-60dB"), (1e-9, 1e-6, "-60..-90dB"), (0.0, 1e-9, "below-90dB")):
        m = (level >= lo) & (level < hi)
        stats[key] = (round(float(m.mean()), 4), int(ulp[m].max()) if m.any() else 0, round(float((ulp[m] > 0).mean()), 4) if m.any() else 0.0)
    print("relinked CLI sp, float32 ulps (%s): %s" % (name, stats))
    # measured (round 5, these two files): within 60 dB of the frame's peak at most ONE ulp apart and 0.1 % of the values
    # differ at all; 60-90 dB down up to 47 ulps (a quarter of them differ); below 90 dB (0.7 % / 4.5 % of the bins) up
    # to 270 / 1143 ulps = 7e-5 relative.  Bounds with room for other boxes' rounding, none for a regression:
    assert stats["0..-30dB"][1] <= 1 and stats["-30..-60dB"][1] <= 1, stats
    assert stats["0..-30dB"][2] < 5e-3 and stats["-30..-60dB"][2] < 5e-3, stats
    assert stats["-60..-90dB"][1] <= 128 and stats["below-90dB"][1] <= 4096, stats
    assert stats["differ"] < 0.15 and stats["more_than_one_ulp"] < 0.1, stats
    ulp_ap = np.abs(gap.view(np.int32).astype(np.int64) - rap.view(np.int32).astype(np.int64))
    assert int(ulp_ap.max()) <= 2 and float((ulp_ap > 0).mean()) < 1e-2, (int(ulp_ap.max()), float((ulp_ap > 0).mean()))
    rl, rm, rb = analysis_files(a_ref, wav, tmp_path, "refc", (5, F, 50, 25))
    gl, gm, gb = analysis_files(a_gpu, wav, tmp_path, "gpuc", (5, F, 50, 25))
    np.testing.assert_allclose(gl, rl, atol=1e-6, rtol=0)
    np.testing.assert_allclose(gm, rm, atol=2e-6, rtol=0)
    np.testing.assert_allclose(gb, rb, atol=2e-6, rtol=0)
    names = [tmp_path / f"ref.{e}" for e in ("f0", "sp", "ap")]
    run(s_ref, *names, tmp_path / "ref_y.wav", 5, F, fs)
    run(s_gpu, *names, tmp_path / "gpu_y.wav", 5, F, fs)
    yr, fr = read_wav(tmp_path / "ref_y.wav")
    yg, fg = read_wav(tmp_path / "gpu_y.wav")
    assert fr == fg == fs and yr.shape == yg.shape
    assert np.abs(yr - yg).max() <= 1 and (yr != yg).mean() < 1e-3
