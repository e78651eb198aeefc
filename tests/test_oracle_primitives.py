"""The oracle's primitives against the golden vectors generated from the compiled reference
(tests/golden/primitives.npz, made by oracle/gen_golden.py) and against numpy where numpy is an
independent authority (FFT conventions)."""
import os

import numpy as np

from conftest import GOLDEN


def gold():
    return np.load(os.path.join(GOLDEN, "primitives.npz"))


def test_randn_table_known_answer(oracle):
    g = gold()
    np.testing.assert_array_equal(oracle.randn_table(4096), g["randn4096"])
    u = oracle.randn_table_u32(4096)
    np.testing.assert_array_equal(u / 268435456.0 - 6.0, g["randn4096"])
    # first value by hand: xorshift128 from the reference seed (matlabfunctions.cpp:247-277)
    x, y, z, w = 123456789, 362436069, 521288629, 88675123
    acc = 0
    for _ in range(12):
        t = (x ^ (x << 11)) & 0xFFFFFFFF
        x, y, z = y, z, w
        w = ((w ^ (w >> 19)) ^ (t ^ (t >> 8))) & 0xFFFFFFFF
        acc = (acc + (w >> 4)) & 0xFFFFFFFF
    assert u[0] == acc


def test_interp1_matches_reference_vectors(oracle):
    g = gold()
    np.testing.assert_array_equal(oracle.interp1(g["interp1_x"], g["interp1_y"], g["interp1_xi"]), g["interp1_out"])


def test_interp1_extrapolates_and_clamps(oracle):
    x = np.array([0.0, 1.0, 3.0])
    y = np.array([0.0, 2.0, 2.0])
    out = oracle.interp1(x, y, np.array([-1.0, 0.0, 0.5, 1.0, 2.0, 3.0, 4.0]))
    np.testing.assert_allclose(out, [-2.0, 0.0, 1.0, 2.0, 2.0, 2.0, 2.0])


def test_decimate(oracle):
    g = gold()
    for r in (2, 3, 6, 12):
        np.testing.assert_array_equal(oracle.decimate(g["decimate_in"], r), g["decimate_%d" % r])


def test_spectral_helpers(oracle):
    g = gold()
    np.testing.assert_array_equal(oracle.dc_correction(g["spec"], 150.0, 16000, 1024), g["dc_150"])
    np.testing.assert_array_equal(oracle.linear_smoothing(g["spec"], 100.0, 16000, 1024), g["ls_100"])
    np.testing.assert_array_equal(oracle.nuttall(769), g["nuttall_769"])


def test_round_and_fft_size(oracle):
    g = gold()
    assert [oracle.lib.orc_matlab_round(float(v)) for v in g["round_in"]] == list(g["round_out"])
    assert [oracle.lib.orc_suitable_fft_size(int(v)) for v in g["fftsize_in"]] == list(g["fftsize_out"])
    assert oracle.lib.orc_suitable_fft_size(1024) == 2048       # 2^k -> 2^(k+1) quirk (common.cpp:51-54)


def test_fft_conventions_vs_numpy(oracle):
    rng = np.random.default_rng(1)
    for n in (8, 64, 1024, 4096):
        x = rng.standard_normal(n)
        spec = oracle.fft_r2c(x)
        np.testing.assert_allclose(spec, np.fft.rfft(x), atol=1e-11)
        assert spec[0].imag == 0.0 and spec[-1].imag == 0.0
        # c2r: unnormalised, ignores Im(DC)/Im(Nyquist) (fft.cpp:27-35)
        dirty = spec.copy()
        dirty[0] += 3j
        dirty[-1] -= 2j
        np.testing.assert_allclose(oracle.fft_c2r(dirty, n), x * n, atol=1e-9)


def test_minimum_phase_vs_numpy_homomorphic(oracle):
    """common.cpp:182-220 is the folded-cepstrum construction; redo it with numpy's FFT."""
    rng = np.random.default_rng(2)
    n = 1024
    ls = np.log(np.abs(rng.standard_normal(n // 2 + 1)) + 0.5)
    mp = oracle.min_phase(ls, n)
    np.testing.assert_allclose(np.abs(mp), np.exp(ls), rtol=1e-12)     # magnitude preserved
    full = np.concatenate([ls, ls[-2:0:-1]])
    cep = np.fft.ifft(full).real
    fold = np.zeros(n)
    fold[0], fold[n // 2] = cep[0], cep[n // 2]
    fold[1:n // 2] = 2 * cep[1:n // 2]
    np.testing.assert_allclose(mp, np.exp(np.fft.fft(fold))[: n // 2 + 1], rtol=1e-10, atol=1e-12)
