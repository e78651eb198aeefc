"""the vibrato feature (data/scripts/Extract.py:115-227).

PARITY UNPINNED: the reference holds no fixtures for it and its LOWESS comes from statsmodels, which is neither
installed here nor pinned by the reference.  So the CPU tests pin the oracle's restatement to properties that
follow from the text of the script and from the published LOWESS algorithm, and the GPU tests compare
hts-train-world_amd/csrc/vibrato.hip with that restatement."""
import importlib
import os

import numpy as np
import pytest

pkg = importlib.import_module("hts-train-world_amd")


def synth_lf0(T, seed, notes):
    """A sung line: per note a pitch with vibrato (5.5 Hz, +-2 %) and drift, unvoiced gaps between some notes."""
    rng = np.random.default_rng(seed)
    f0 = np.zeros(T)
    segs, pos = [], 0
    for k, (name, frames, gap) in enumerate(notes):
        hz = pkg.recipe.note_pitch(name)
        n = min(frames, T - pos)
        if n <= 0:
            break
        t = np.arange(n)
        depth = 0.02 * hz * (1.0 + 0.5 * np.sin(t / 40.0 + k))
        f0[pos:pos + n] = hz * (1.0 + 0.01 * np.sin(t / 90.0)) + depth * np.sin(2 * np.pi * 5.5 * t * 0.005 + k) + rng.normal(0, 0.3, n)
        if gap:
            f0[pos + n - gap:pos + n] = 0.0                       # closes the run inside the segment
        segs.append((pos, pos + n, hz))
        pos += n
    lf0 = np.where(f0 > 0, np.log(np.maximum(f0, 1e-9)), 0.0).astype(np.float32)
    return lf0, segs


NOTES = [("A3", 180, 6), ("C4", 90, 4), ("E4", 240, 0), ("D4", 30, 3), ("xx", 60, 0), ("G3", 300, 10), ("A3", 150, 5)]


def test_lowess_properties(oracle):
    x = np.arange(80.0)
    lin = 0.7 * x - 11.0
    np.testing.assert_allclose(oracle.lowess(lin), lin, atol=1e-10)          # a local LINEAR fit reproduces a line
    np.testing.assert_allclose(oracle.lowess(np.full(33, 4.25)), 4.25, atol=1e-12)
    smooth = 30.0 * np.sin(x / 25.0)
    clean = oracle.lowess(smooth)
    dirty = smooth.copy()
    dirty[40] += 1000.0
    # 20 robustifying passes give the outlier weight zero: the fit is the clean one again away from the window edge
    assert np.abs(oracle.lowess(dirty) - clean).max() < 0.5
    assert np.abs(oracle.lowess(dirty, it=0) - clean).max() > 10.0
    # span: k = int(2/3 n) neighbours; a bump narrower than the span is smoothed away, a trend is kept
    bump = np.zeros(90)
    bump[44:47] = 9.0
    assert np.abs(oracle.lowess(bump)).max() < 1.0
    # frac = 1 with it = 0 weights all points by the tricube of their distance: still exact on lines
    np.testing.assert_allclose(oracle.lowess(lin, frac=1.0, it=0), lin, atol=1e-10)


def test_get_vibrate_reads_like_the_script(oracle):
    n = 200
    f = 30.0 * np.sin(np.arange(n) / 5.0)                        # crossings at 16, 32, 48, ... (every 5 pi frames)
    t = oracle.get_vibrate(f)
    assert np.all(t[:n] == 0)                                    # the list starts as `length` zero pairs (:118)
    ip = [i for i in range(1, n) if (f[i - 1] > 0) != (f[i] > 0)]
    assert len(t) == n + (n - ip[0])                             # every frame from the first crossing on is appended
    first = t[n:n + ip[1] - ip[0]]
    np.testing.assert_allclose(first[:, 0], np.abs(f[ip[0]:ip[1]]).max())
    assert np.all(first[:, 1] == ip[1] - ip[0] / 2.0)            # period = end - start / 2 (:147)
    # a half-wave below 5 is skipped and shifts what follows
    g = f.copy()
    g[ip[2]:ip[3]] *= 0.1
    t2 = oracle.get_vibrate(g)
    assert len(t2) == len(t) - (ip[3] - ip[2])
    # no crossing at all: nothing appended (the script would raise)
    assert len(oracle.get_vibrate(np.full(50, 3.0))) == 50


def test_run_selection_and_spill_over(oracle):
    lf0, segs = synth_lf0(1100, 3, NOTES)
    vib, lf2, runs = oracle.vibrato(lf0, segs)
    # runs: notes 0, 1, 5, 6 are closed by a gap inside their segment and longer than 20 frames; note 2 reaches
    # the end of its segment voiced (never processed); note 3 is too short; note 4 ("xx") is voiced but continues
    # note 2's stretch only inside its own segment
    assert runs >= 4
    s0, e0, _ = segs[0]
    # soprLog writes the literal 1e-8 for values <= 0 (not its logarithm, Extract.py:91-92)
    assert np.all(vib[s0:e0 - 6] == np.float32(1e-8))            # the run's own frames are zeroed (:223-225)
    after = vib[e0 - 6:e0 - 6 + 100, 0].astype(np.float64)
    assert np.any(np.exp(after[after != np.float32(1e-8)]) > 4.9)   # its pairs lie AFTER it (depth >= 5)
    # second column of the rewritten lf0: log(f0 - note + 500) on voiced frames, 1e-8 elsewhere
    f0 = np.where(np.exp(lf0.astype(np.float64)) < 1, 0, np.exp(lf0.astype(np.float64)))
    v = (f0 >= 55) & (np.arange(len(f0)) < segs[-1][1])
    hz = np.zeros(len(f0))
    for a, b, p in segs:
        hz[a:b] = p
    np.testing.assert_allclose(lf2[v, 1], np.log(f0[v] - hz[v] + 500.0).astype(np.float32), rtol=1e-6)
    np.testing.assert_allclose(lf2[v, 0], np.log(f0[v]).astype(np.float32), rtol=1e-6)


def test_label_reading(tmp_path):
    recipe = pkg.recipe
    assert abs(recipe.note_pitch("A4") - 440.0) < 1e-9 and abs(recipe.note_pitch("A3") - 220.0) < 1e-9
    assert abs(recipe.note_pitch("C4") - 261.6255653) < 1e-6 and recipe.note_pitch("xx") == 0.0
    (tmp_path / "m.lab").write_text("0 5000000 sil\n5000000 12500000 a\n12500000 20000000 i\n")
    (tmp_path / "f.lab").write_text("0 5000000 x^sil-a+i/E:xx]1\n5000000 12500000 sil^a-i+u/E:A4]2\n"
                                    "12500000 20000000 a^i-u+e/E:Db5]3\n")
    segs = recipe.read_label_segments(tmp_path / "m.lab", tmp_path / "f.lab", 5.0, 390)
    assert [(a, b) for a, b, _ in segs] == [(0, 100), (100, 250), (250, 390)]        # 100 ns units / 10e3 = ms
    assert segs[0][2] == 0.0 and abs(segs[1][2] - 440.0) < 1e-9 and abs(segs[2][2] - 554.365262) < 1e-5


@pytest.mark.gpu
def test_vibrato_on_gpu_against_the_restatement(gpu, oracle):
    torch, W, ctx = gpu
    utts = [synth_lf0(1100, 3, NOTES), synth_lf0(700, 4, NOTES[2:] + NOTES[:2]), synth_lf0(40, 5, NOTES[:1]),
            synth_lf0(1500, 6, [("G3", 900, 8), ("A3", 600, 7)])]
    b = W.WorldBatch(ctx, W.default_params(48000, 5.0), f0_lengths=[len(u[0]) for u in utts])
    lf0 = torch.from_numpy(np.concatenate([u[0] for u in utts])).cuda()
    vib, lf2, too_long = b.vibrato(lf0, [u[1] for u in utts])
    assert too_long == 0
    vparts, lparts = b.split_frames(vib.cpu().numpy()), b.split_frames(lf2.cpu().numpy())
    total_runs = 0
    for (l, segs), gv, gl in zip(utts, vparts, lparts):
        wv, wl, runs = oracle.vibrato(l, segs)
        total_runs += runs
        np.testing.assert_allclose(gl, wl, rtol=1e-6, atol=0)
        np.testing.assert_allclose(gv, wv, rtol=2e-6, atol=0)
    assert total_runs >= 8
    b.close()


@pytest.mark.gpu
def test_vibrato_files(gpu, oracle, tmp_path):
    torch, W, ctx = gpu
    recipe = pkg.recipe
    lf0, segs = synth_lf0(1100, 7, NOTES)
    lf0.tofile(tmp_path / "a.lf0")
    inv = {round(recipe.note_pitch(n), 6): n for n in ("A3", "C4", "E4", "D4", "G3")}
    with open(tmp_path / "a.mono", "w") as m, open(tmp_path / "a.full", "w") as f:
        for a, e, hz in segs:
            note = inv.get(round(hz, 6), "xx")
            m.write("%d %d la\n" % (a * 50000, e * 50000))          # 5 ms frames in 100 ns units
            f.write("%d %d x^la-y/E:%s]q\n" % (a * 50000, e * 50000, note))
    n = recipe.vibrato_files([(tmp_path / "a.lf0", tmp_path / "a.mono", tmp_path / "a.full", tmp_path / "a.vib")], 5.0,
                             ctx=ctx)
    assert n == 1100
    wv, wl, _ = oracle.vibrato(lf0, segs)
    np.testing.assert_allclose(np.fromfile(tmp_path / "a.vib", dtype=np.float32).reshape(-1, 2), wv, rtol=2e-6)
    np.testing.assert_allclose(np.fromfile(tmp_path / "a.lf0", dtype=np.float32).reshape(-1, 2), wl, rtol=1e-6)
