import numpy as np
rng = np.random.default_rng(0)
H = 512; F = 2 * H
def ref_minphase(ls):
    full = np.concatenate([ls, ls[-2:0:-1]])           # length F, even
    C = np.fft.fft(full).real
    c = np.zeros(F); c[0] = C[0]; c[1:H] = 2 * C[1:H]; c[H] = C[H]
    S = np.fft.fft(c)                                    # forward, as the current kernel does
    return C, S[:H + 1]
def even_dft_pair(w):
    """w: complex [H+1]; returns DFT_F of its even extension at j = 0..H (complex: Cp + i Ca)"""
    full = np.concatenate([w, w[-2:0:-1]])
    z = full[0::2] + 1j * full[1::2]                     # e + i o, length H
    Z = np.fft.fft(z)
    j = np.arange(H + 1)
    Zj = Z[j % H]; Zm = Z[(H - j) % H]
    out = np.zeros(H + 1, complex)
    # j = 0 and j = H: sin = 0 -> handle directly: A[0] = sum(full), A[H] = sum(full * (-1)^n)
    out[0] = full.sum(); out[H] = (full[0::2].sum() - full[1::2].sum())
    jj = np.arange(1, H)
    R = -(Zj[jj] - Zm[jj]) / (2 * np.sin(np.pi * jj / H))
    E = Zj[jj] - 1j * np.exp(1j * np.pi * jj / H) * R
    out[jj] = E + R
    return out
def odd_dft_pair(g):
    """g: complex [H+1] with g[0] = g[H] = 0; returns DFT_F of its odd extension at k = 0..H"""
    full = np.concatenate([g, -g[-2:0:-1]])
    z = full[0::2] + 1j * full[1::2]
    Z = np.fft.fft(z)
    k = np.arange(H + 1)
    Zk = Z[k % H]; Zm = Z[(H - k) % H]
    out = np.zeros(H + 1, complex)
    kk = np.arange(1, H)
    R = -(Zk[kk] + Zm[kk]) / (2 * np.sin(np.pi * kk / H))
    E = Zk[kk] - 1j * np.exp(1j * np.pi * kk / H) * R
    out[kk] = E + R
    return out
# realistic log spectra: smooth envelope in log domain
k = np.arange(H + 1)
env = 1e-3 * (1 / (1 + ((k - 40) / 10.0) ** 2) + 0.5 / (1 + ((k - 120) / 15.0) ** 2) + 1e-4) * np.exp(0.3 * rng.standard_normal(H + 1))
rat = np.clip(0.02 + 0.9 * (k / H) ** 2 + 0.02 * rng.standard_normal(H + 1), 0.001, 0.999999999999) ** 2
lp = np.log(env * (1 - rat) + 1e-12) / 2; la = np.log(env * rat) / 2
Cp, Sp = ref_minphase(lp); Ca, Sa = ref_minphase(la)
A = even_dft_pair(lp + 1j * la)
print("cepstrum err p %.2e a %.2e (scale %.1e)" % (np.abs(A.real - Cp[:H+1]).max(), np.abs(A.imag - Ca[:H+1]).max(), np.abs(Cp).max()))
g = np.zeros(H + 1, complex); g[1:H] = 2 * A[1:H]
G = odd_dft_pair(g)        # DFT_F of odd extension = -2i sum c sin -> Im S = real part of (G / -i)... check:
# S[k] = c0 + (-1)^k cH + sum_{0<j<H} c_j e^{-i th} + ... ; Im S[k] = -sum_{0<j<H} c_j sin(pi j k / H) with c_j = 2 C_j
# G[k] = sum_full g e^{-i..} = -2i sum_{0<j<H} g_j sin(pi j k/H) ; g_j = 2 C_j (complex pair)  => G = -2i * sum 2C sin
# Im S_p = -sum 2 Cp sin = Re(G)/(... ) : G = -2i (Xp + i Xa), X = sum 2C sin -> G = 2 Xa - 2i Xp -> Xp = -Im G / 2, Xa = Re G / 2
Xp = -G.imag / 2; Xa = G.real / 2
print("Im S err p %.2e a %.2e (scale %.1e)" % (np.abs(-Xp - Sp.imag).max(), np.abs(-Xa - Sa.imag).max(), np.abs(Sp.imag).max()))
print("Re S/F vs ls: p %.2e a %.2e" % (np.abs(Sp.real / F - lp).max(), np.abs(Sa.real / F - la).max()))
ph_err = np.abs((-Xp - Sp.imag) / F).max()
print("phase err rad %.2e" % ph_err)
