"""Synthetic speech-like utterances for parity tests and bench.py.

The reference ships two wav files and no corpus, and there
is no network, so every workload is generated here from fixed seeds.  The
generator follows the shape laid down in SURVEY.md section 8(d): alternating
voiced / unvoiced segments, a harmonic source under a two-resonance envelope
with drifting + vibrato F0, noise floors so that no sample run is digital
silence, peak-normalised and quantised to int16 so that ``x = s / 32768`` is
exactly what the reference's ``wavread`` would hand to the library
(externs/WORLD_v2/test/audioio.cpp:236-249).

Pure numpy, deterministic (counter-based splitmix64), no global state.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(counter: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser over uint64 counters."""
    with np.errstate(over="ignore"):
        z = (counter + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n doubles in [0, 1) from (seed, stream); counter based, reproducible."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream) * np.uint64(0x9E3779B1)
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    return (_splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _envelope(f: np.ndarray) -> np.ndarray:
    """Two Lorentzian resonances (600/300 Hz, 1800/400 Hz) plus a floor."""
    r1 = 1.0 / (1.0 + ((f - 600.0) / 150.0) ** 2)
    r2 = 0.6 / (1.0 + ((f - 1800.0) / 200.0) ** 2)
    return r1 + r2 + 0.02


def utterance_samples(index: int, fs: int = 16000, dur_range=(2.0, 8.0)) -> int:
    """Sample count of make_utterance(index, fs, dur_range) without generating it (the sweep's plan needs
    only the lengths, like a wav header)."""
    p0 = uniform(1000 + int(index), 0, 64)[0]
    return int(round((dur_range[0] + (dur_range[1] - dur_range[0]) * p0) * fs))


def make_utterance(index: int, fs: int = 16000, dur_range=(2.0, 8.0),
                   duration: float | None = None) -> np.ndarray:
    """Utterance ``index`` (seed 1000+index) as float64 samples s/32768."""
    seed = 1000 + int(index)
    p = uniform(seed, 0, 64)
    dur = float(duration) if duration is not None else dur_range[0] + (dur_range[1] - dur_range[0]) * p[0]
    n = int(round(dur * fs))
    t = np.arange(n) / fs
    base = 90.0 + 170.0 * p[1]
    drift_hz = 0.3 + 0.7 * p[2]
    f0 = base * (1.0 + 0.25 * np.sin(2 * np.pi * (drift_hz * t + p[3]))
                 + 0.03 * np.sin(2 * np.pi * (5.5 * t + p[4])))
    # voiced/unvoiced segmentation
    voiced = np.zeros(n)
    pos, k, is_v = 0.0, 8, p[5] < 0.7
    while pos < dur and k < 62:
        length = (0.3 + 0.9 * p[k]) if is_v else (0.1 + 0.3 * p[k])
        a, b = int(pos * fs), min(n, int((pos + length) * fs))
        if is_v:
            voiced[a:b] = 1.0
        pos += length
        k += 1
        is_v = not is_v
    ramp = int(0.010 * fs)
    if ramp > 1:                          # 10 ms raised-cosine edges on the voicing gate
        win = np.hanning(2 * ramp + 1)
        win /= win.sum()
        voiced = np.convolve(voiced, win, mode="same")
    phase = 2 * np.pi * np.cumsum(f0) / fs
    src = np.zeros(n)
    hmax = int(0.45 * fs / f0.min())
    for h in range(1, hmax + 1):
        fh = h * f0
        src += np.where(fh < 0.45 * fs, _envelope(fh) * np.sin(h * phase) / h, 0.0)
    src /= max(1e-9, np.abs(src).max())
    noise = 2.0 * uniform(seed, 1, n) - 1.0
    x = voiced * (src + 10 ** (-50 / 20) * noise / 0.3) + (1.0 - voiced) * (10 ** (-30 / 20) * noise / 0.3)
    x *= 0.3 / np.abs(x).max()
    s = np.round(x * 32768.0).clip(-32768, 32767).astype(np.int16)
    return s.astype(np.float64) / 32768.0


def make_batch(count: int, fs: int = 16000, dur_range=(2.0, 8.0), first: int = 0,
               workers: int = 1, indices=None) -> list[np.ndarray]:
    """``count`` utterances with indices first..first+count-1 (or the given ``indices``)."""
    idx = list(indices) if indices is not None else range(first, first + count)
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            return pool.starmap(make_utterance, [(i, fs, dur_range) for i in idx])
    return [make_utterance(i, fs, dur_range) for i in idx]
