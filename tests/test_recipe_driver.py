"""The batched file driver (hts-train-world_amd/recipe.py) against the reference's own CLIs run per utterance
(oracle/_ref/cli/analysis_ref, synth_ref: the recipe's loop, data/Makefile.in:206-216) -- file for file."""
import importlib
import os
import struct
import subprocess

import numpy as np
import pytest

pkg = importlib.import_module("hts-train-world_amd")
sd, recipe = pkg.synth_data, pkg.recipe
from test_cli_relink import cli, run, write_wav as write_wav_py  # noqa: E402


def test_wav_formats(tmp_path):
    """test/audioio.cpp: wavwrite truncates y * 32767 toward zero and clips; wavread divides by 2^(nbit-1)."""
    y = np.array([0.0, 0.5, -0.5, 0.99999, -0.99999, 1.5, -1.5, 3.05e-5, -3.05e-5, 6.2e-5])
    recipe.write_wav(tmp_path / "a.wav", y, 16000)
    raw = open(tmp_path / "a.wav", "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[36:40] == b"data" and len(raw) == 44 + 2 * len(y)
    assert struct.unpack("<IHHIIHH", raw[16:36]) == (16, 1, 1, 16000, 32000, 2, 16)
    s = np.frombuffer(raw[44:], dtype="<i2")
    np.testing.assert_array_equal(s, [0, 16383, -16383, 32766, -32766, 32767, -32768, 0, 0, 2])
    x, fs = recipe.read_wav(tmp_path / "a.wav")
    assert fs == 16000
    np.testing.assert_array_equal(x, s / 32768.0)
    # 24-bit PCM
    vals = np.array([0, 1, -1, 8388607, -8388608, 123456], dtype=np.int64)
    body = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in vals)
    head = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 8000, 24000, 3, 24)
    open(tmp_path / "b.wav", "wb").write(head + b"data" + struct.pack("<I", len(body)) + body)
    x, fs = recipe.read_wav(tmp_path / "b.wav")
    np.testing.assert_array_equal(x, vals / 8388608.0)


def test_sharing_of_jobs(monkeypatch):
    costs = [50, 10, 40, 30, 20, 60]
    assert recipe._my_share(costs) == list(range(6))
    seen = []
    for r in range(2):
        monkeypatch.setenv("WORLD_SIZE", "2")
        monkeypatch.setenv("RANK", str(r))
        seen.append(recipe._my_share(costs))
    assert sorted(seen[0] + seen[1]) == list(range(6))
    assert abs(sum(costs[i] for i in seen[0]) - sum(costs[i] for i in seen[1])) <= 10
    groups = list(recipe._batches([5, 0, 2, 3, 4, 1], costs, 100))
    assert [i for g in groups for i in g] == [5, 0, 2, 3, 4, 1] and all(sum(costs[i] for i in g) <= 100 for g in groups)


def test_outputs_complete_is_a_size_check(tmp_path):
    """`resume` skips an utterance only when its three files have exactly the sizes its analysis writes."""
    n, fs = 137, 16000                                        # frames, CheapTrick's fft 1024 -> 513 bins
    job = ("x.wav", tmp_path / "a.f0", tmp_path / "a.sp", tmp_path / "a.ap")
    assert not recipe.outputs_complete(job, n, fs)            # nothing there
    (tmp_path / "a.f0").write_bytes(b"\0" * (4 * n))
    (tmp_path / "a.sp").write_bytes(b"\0" * (4 * n * 513))
    (tmp_path / "a.ap").write_bytes(b"\0" * (4 * n * 513 - 4))         # cut short
    assert not recipe.outputs_complete(job, n, fs)
    (tmp_path / "a.ap").write_bytes(b"\0" * (4 * n * 513))
    assert recipe.outputs_complete(job, n, fs)
    assert not recipe.outputs_complete(job, n, fs, spec_dim=50, ap_dim=25)     # raw files are not the coded ones
    (tmp_path / "a.sp").write_bytes(b"\0" * (4 * n * 50))
    (tmp_path / "a.ap").write_bytes(b"\0" * (4 * n * 25))
    assert recipe.outputs_complete(job, n, fs, spec_dim=50, ap_dim=25)
    assert not recipe.outputs_complete(job, n + 1, fs, spec_dim=50, ap_dim=25)


@pytest.mark.gpu
def test_driver_resumes_a_run_that_was_cut_short(gpu, tmp_path):
    """analysis_files(resume=True): utterances whose files are complete are left alone (same bytes, same mtime), a
    missing and a truncated one are analysed again -- and come out as in the uninterrupted run."""
    specs = [(16000, 45, 0.6), (16000, 46, 0.9), (16000, 47, 0.4), (48000, 48, 0.5)]
    jobs = []
    for k, (fs, idx, dur) in enumerate(specs):
        wav = tmp_path / f"r{k}.wav"
        write_wav_py(wav, sd.make_utterance(idx, fs, duration=dur), fs)
        jobs.append((wav, tmp_path / f"r{k}.lf0", tmp_path / f"r{k}.mgc", tmp_path / f"r{k}.bap"))
    n_all = recipe.analysis_files(jobs, 5.0, 0, 50, 25)
    want = {p: open(p, "rb").read() for j in jobs for p in j[1:]}
    os.remove(jobs[1][2])                                     # one file of utterance 1 missing
    with open(jobs[3][3], "r+b") as f:                        # utterance 3's last file cut short
        f.truncate(os.path.getsize(jobs[3][3]) - 100)
    before = {p: os.stat(p).st_mtime_ns for j in (jobs[0], jobs[2]) for p in j[1:]}
    n_again = recipe.analysis_files(jobs, 5.0, 0, 50, 25, resume=True)
    assert 0 < n_again < n_all
    for p, data in want.items():
        assert open(p, "rb").read() == data, p
    for p, t in before.items():
        assert os.stat(p).st_mtime_ns == t, p                 # untouched
    assert recipe.analysis_files(jobs, 5.0, 0, 50, 25, resume=True) == 0
    with pytest.raises(ValueError):
        recipe.analysis_files(jobs, 5.0, 0, 50, 25, resume=True, gather=True)


@pytest.mark.gpu
@pytest.mark.parametrize("coded", [False, True])
def test_driver_writes_the_clis_files(gpu, tmp_path, coded):
    a_ref, s_ref = cli("analysis_ref"), cli("synth_ref")
    specs = [(16000, 41, 0.9), (16000, 42, 1.7), (48000, 43, 0.8), (16000, 44, 0.25)]
    extra = (50, 25) if coded else ()
    jobs, refs = [], []
    for k, (fs, idx, dur) in enumerate(specs):
        wav = tmp_path / f"u{k}.wav"
        write_wav_py(wav, sd.make_utterance(idx, fs, duration=dur), fs)
        jobs.append((wav, tmp_path / f"u{k}.f0", tmp_path / f"u{k}.sp", tmp_path / f"u{k}.ap"))
        refs.append((tmp_path / f"r{k}.f0", tmp_path / f"r{k}.sp", tmp_path / f"r{k}.ap"))
        run(a_ref, wav, *refs[-1], 5, 0, *extra)                       # fft size 0: CheapTrick's own
    # tiny batches on purpose: several launches, mixed sampling rates
    n = recipe.analysis_files(jobs, 5.0, 0, *(extra or (0,)), ctx=gpu[2], max_batch_frames=250)
    assert n == sum(os.path.getsize(r[0]) // 4 for r in refs)
    for job, ref in zip(jobs, refs):
        for got, want, tol in zip(job[1:], ref, (1e-6, 2e-6, 2e-6) if coded else (0.0, 0.0, 0.0)):
            g, w = np.fromfile(got, dtype=np.float32), np.fromfile(want, dtype=np.float32)
            assert g.shape == w.shape
            if coded:
                np.testing.assert_allclose(g, w, atol=tol, rtol=0)
            else:
                assert (g != w).mean() < 1e-2                              # one float32 ulp at rounding boundaries
                np.testing.assert_allclose(g, w, rtol=3e-7, atol=1e-12)
    if coded:
        return            # the reference's coded synth reads uninitialised ap bins: nothing to compare file-wise
    # synth from the reference's files, per sampling rate (one call of the CLI fixes fs and fft size)
    for fs, F in ((16000, 1024), (48000, 2048)):
        sel = [k for k, s in enumerate(specs) if s[0] == fs]
        sj = [(*refs[k], tmp_path / f"u{k}_y.wav") for k in sel]
        recipe.synth_files(sj, 5.0, F, fs, ctx=gpu[2], max_batch_frames=250)
        for k, j in zip(sel, sj):
            run(s_ref, *refs[k], tmp_path / f"r{k}_y.wav", 5, F, fs)
            a, b = open(j[3], "rb").read(), open(tmp_path / f"r{k}_y.wav", "rb").read()
            assert a[:44] == b[:44] and len(a) == len(b)
            ya, yb = np.frombuffer(a[44:], dtype="<i2").astype(int), np.frombuffer(b[44:], dtype="<i2").astype(int)
            assert np.abs(ya - yb).max() <= 1 and (ya != yb).mean() < 1e-3


@pytest.mark.gpu
def test_driver_coded_synthesis(gpu, oracle, tmp_path):
    """synth in the recipe's coded form against the oracle's restatement of synth.cpp:151-256 + Synthesis."""
    fs, F = 16000, 1024
    wav = tmp_path / "in.wav"
    write_wav_py(wav, sd.make_utterance(45, fs, duration=1.2), fs)
    job = (wav, tmp_path / "a.lf0", tmp_path / "a.mgc", tmp_path / "a.bap")
    recipe.analysis_files([job], 5.0, F, 50, 25, ctx=gpu[2])
    recipe.synth_files([(*job[1:], tmp_path / "y.wav")], 5.0, F, fs, 50, 25, ctx=gpu[2])
    lf0 = np.fromfile(job[1], dtype=np.float32)
    mgc = np.fromfile(job[2], dtype=np.float32).reshape(-1, 50)
    bap = np.fromfile(job[3], dtype=np.float32).reshape(-1, 25)
    f0, sp, ap = oracle.recipe_decode(lf0, mgc, bap, fs, F)
    y = oracle.synthesis(f0, sp, ap, F, 5.0, fs)
    want = np.clip(np.trunc(y * 32767.0), -32768, 32767).astype(int)
    got, gfs = recipe.read_wav(tmp_path / "y.wav")
    got = np.round(got * 32768).astype(int)
    assert gfs == fs and got.shape == want.shape
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 1e-3


@pytest.mark.gpu
def test_driver_cmp_files(gpu, tmp_path):
    """cmp stage for a file list against the Perl scripts' own outputs (tests/golden/cmp_windows.npz)."""
    from test_golden import CMP_WINDOWS, GOLDEN, cmp_stream
    g = np.load(os.path.join(GOLDEN, "cmp_windows.npz"))
    names = ("mgc", "lf0", "bap")
    T = int(g["mgc_shape"][0])
    for k in range(2):                                     # the same utterance twice: two jobs, one batch
        for n in names:
            cmp_stream(int(g[n + "_seed"]), T, int(g[n + "_shape"][1]), bool(g[n + "_holes"])).tofile(tmp_path / f"{k}.{n}")
    for n, w in zip(names, CMP_WINDOWS):                   # window files in the recipe's text form (data/win)
        for i, c in enumerate(CMP_WINDOWS, 1):
            open(tmp_path / f"{n}.win{i}", "w").write("%d %s\n" % (len(c), " ".join(repr(v) for v in c)))
    jobs = [tuple(tmp_path / f"{k}.{n}" for n in names) + (tmp_path / f"{k}.cmp",) for k in range(2)]
    streams = [(int(g[n + "_shape"][1]), [tmp_path / f"{n}.win{i}" for i in (1, 2, 3)]) for n in names]
    sr, shift, byte, kind = (int(v) for v in g["htk_args"])
    assert recipe.cmp_files(jobs, streams, sr, shift, kind, ctx=gpu[2]) == 2 * T
    want = bytes(g["htk_header"]) + np.concatenate([g[n + "_windowed"] for n in names], axis=1).tobytes()
    for k in range(2):
        assert open(tmp_path / f"{k}.cmp", "rb").read() == want


@pytest.mark.gpu
def test_driver_command_line(gpu, tmp_path):
    """python -m hts-train-world_amd.recipe analysis / synth over a job list, as the recipe would call it."""
    import sys
    fs, F = 16000, 1024
    lines_a, lines_s = [], []
    for k, dur in enumerate((0.5, 0.8)):
        wav = tmp_path / f"c{k}.wav"
        write_wav_py(wav, sd.make_utterance(50 + k, fs, duration=dur), fs)
        lines_a.append(f"{wav} {tmp_path}/c{k}.lf0 {tmp_path}/c{k}.mgc {tmp_path}/c{k}.bap")
        lines_s.append(f"{tmp_path}/c{k}.lf0 {tmp_path}/c{k}.mgc {tmp_path}/c{k}.bap {tmp_path}/c{k}_y.wav")
    (tmp_path / "a.scp").write_text("\n".join(lines_a) + "\n")
    (tmp_path / "s.scp").write_text("# comment\n" + "\n".join(lines_s) + "\n")
    root = os.path.join(os.path.dirname(__file__), "..")
    base = [sys.executable, "-m", "hts-train-world_amd.recipe"]
    r = subprocess.run(base + ["analysis", "--scp", str(tmp_path / "a.scp"), "--frame-period", "5", "--fft-size", str(F),
                               "--spec-dim", "50", "--ap-dim", "25"], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "complete." in r.stdout, r.stdout + r.stderr
    r = subprocess.run(base + ["synth", "--scp", str(tmp_path / "s.scp"), "--frame-period", "5", "--fft-size", str(F),
                               "--fs", str(fs), "--spec-dim", "50", "--ap-dim", "25"], cwd=root, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "complete." in r.stdout, r.stdout + r.stderr
    for k, dur in enumerate((0.5, 0.8)):
        T = os.path.getsize(tmp_path / f"c{k}.lf0") // 4
        assert os.path.getsize(tmp_path / f"c{k}.mgc") == T * 50 * 4 and os.path.getsize(tmp_path / f"c{k}.bap") == T * 25 * 4
        y, yfs = recipe.read_wav(tmp_path / f"c{k}_y.wav")
        assert yfs == fs and len(y) == int((T - 1) * 5.0 / 1000.0 * fs) + 1 and np.abs(y).max() > 0.01
