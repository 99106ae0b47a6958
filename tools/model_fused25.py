"""NumPy model of the data flow of k_fused25 (detprocess_amd/csrc/ofx_fused25.hip): index maps of
the three register stages, the pair middle step and its tables, checked against numpy.fft.
Development aid (executable specification of the kernel's geometry); not used by the product."""
import numpy as np

R1, R2, R3 = 20, 25, 25
M = R1 * R2 * R3
N = 2 * M
P = R1 * R2            # blocks of R3 bins
FT = P // 2            # pair virtual threads = hardware threads in use


def dft25(x, sign):
    # 5 x 5 Cooley-Tukey as in ofx_fft_mixed.h (positions, twiddles, renaming)
    x = x.copy()
    w5 = lambda j: np.exp(sign * 2j * np.pi * j / 5)
    def r5(idx):
        v = x[idx].copy()
        for k in range(5):
            x[idx[k]] = sum(v[j] * w5(j * k) for j in range(5))
    for b in range(5):
        r5([5 * a + b for a in range(5)])
    for i in range(25):
        c, b = divmod(i, 5)
        x[i] *= np.exp(sign * 2j * np.pi * b * c / 25)
    for c in range(5):
        r5([5 * c + b for b in range(5)])
    return np.array([x[5 * (k % 5) + k // 5] for k in range(25)])


def dft20(x, sign):
    x = x.copy()
    p20 = lambda a, b: (5 * a + 4 * b) % 20
    for n2 in range(5):
        idx = [p20(n1, n2) for n1 in range(4)]
        v = x[idx].copy()
        for k in range(4):
            x[idx[k]] = sum(v[j] * np.exp(sign * 2j * np.pi * j * k / 4) for j in range(4))
    for k1 in range(4):
        idx = [p20(k1, n2) for n2 in range(5)]
        v = x[idx].copy()
        for k in range(5):
            x[idx[k]] = sum(v[j] * np.exp(sign * 2j * np.pi * j * k / 5) for j in range(5))
    return np.array([x[p20(k % 4, k % 5)] for k in range(20)])


def forward(z):
    """z[M] -> Zblk[k_low][k3] with Z_k at k = k_low + P k3,  k_low = k1 + R1 k2."""
    # F1: vt n' = R3 n2 + n3 holds z[(R2 R3) n1 + n'], radix-20 over n1, times w_M^{n' k1}
    a = np.zeros((R2 * R3, R1), complex)
    for n_ in range(R2 * R3):
        a[n_] = dft20(z[R2 * R3 * np.arange(R1) + n_], -1) * np.exp(-2j * np.pi * n_ * np.arange(R1) / M)
    # E1 + F2: vt (k1, n3) holds n2 = 0..R2-1, radix-25 over n2, times w_{R2 R3}^{n3 k2}
    b = np.zeros((R1, R3, R2), complex)
    for k1 in range(R1):
        for n3 in range(R3):
            v = np.array([a[R3 * n2 + n3, k1] for n2 in range(R2)])
            b[k1, n3] = dft25(v, -1) * np.exp(-2j * np.pi * n3 * np.arange(R2) / (R2 * R3))
    # E2 + F3: block k_low = k1 + R1 k2 holds n3 = 0..R3-1, radix-25 over n3
    Zb = np.zeros((P, R3), complex)
    for k1 in range(R1):
        for k2 in range(R2):
            Zb[k1 + R1 * k2] = dft25(b[k1, :, k2], -1)
    return Zb


def inverse(Zb):
    b = np.zeros((R1, R3, R2), complex)
    for k1 in range(R1):
        for k2 in range(R2):
            b[k1, :, k2] = dft25(Zb[k1 + R1 * k2], +1)
    a = np.zeros((R2 * R3, R1), complex)
    for k1 in range(R1):
        for n3 in range(R3):
            v = dft25(b[k1, n3] * np.exp(+2j * np.pi * n3 * np.arange(R2) / (R2 * R3)), +1)
            for n2 in range(R2):
                a[R3 * n2 + n3, k1] = v[n2]
    z = np.zeros(M, complex)
    for n_ in range(R2 * R3):
        z[R2 * R3 * np.arange(R1) + n_] = dft20(a[n_] * np.exp(+2j * np.pi * n_ * np.arange(R1) / M), +1)
    return z


def pair_bins(v, J):
    """bins (k, p) of slot J of pair thread v (generic layout after perm_in for v = 0)."""
    if v != 0:
        k = v + P * J
    else:
        k = P * J if J <= 12 else P // 2 + P * (J - 13)
    return k, (M - k) % M


def perm_in(A0, B0):
    genA = np.concatenate([A0[:13], B0[:12]])
    genB = np.concatenate([B0[13:25], A0[13:25], A0[:1]])
    return genA, genB


def perm_out(genA, genB, self_val):
    A0 = np.concatenate([genA[:13], genB[12:24]])
    B0 = np.concatenate([genA[13:25], [self_val], genB[:12]])
    return A0, B0


def middle(Zb, wf, g):
    """pair middle step on the blocks; wf[K] one-sided filter (A = irfft-like of wf V), g[K].
    Returns the blocks of Z' (input of the inverse packed transform) and chi2_0."""
    out = np.zeros_like(Zb)
    chi = 0.0
    for v in range(FT):
        A = Zb[v].copy()
        B = Zb[P - v if v else P // 2].copy()
        if v == 0:
            selfv = B[12]
            A, B = perm_in(A, B)
        for J in range(R3):
            k, p = pair_bins(v, J)
            zk, zp = A[J], B[R3 - 1 - J]
            if v == 0 and J == 0:
                zp = zk
            T = 1j * np.exp(-2j * np.pi * k / N)
            if k == 0:
                wk, wp, gk, gp = wf[0] / 2, np.conj(wf[M]) / 2, g[0] / 4, g[M] / 4
            else:
                wk, wp, gk, gp = wf[k] / 2, np.conj(wf[p]) / 2, g[k] / 2, g[p] / 2
            u = zk + np.conj(zp)
            w = zk - np.conj(zp)
            sv = w * T
            xk2, xp2 = u - sv, u + sv
            chi += gk * (xk2.real ** 2 + xk2.imag ** 2) + gp * (xp2.real ** 2 + xp2.imag ** 2)
            yk, yp = xk2 * wk, xp2 * wp
            sg, df = yk + yp, yk - yp
            q = df * np.conj(T)
            A[J] = sg - q
            B[R3 - 1 - J] = np.conj(sg + q)
        if v == 0:
            zq = selfv * np.conj(wf[M // 2])
            chi += 2 * g[M // 2] * (selfv.real ** 2 + selfv.imag ** 2)
            A, B = perm_out(A, B, 2 * zq)
        out[v] = A
        out[P - v if v else P // 2] = B
    return out, chi


if __name__ == '__main__':
    rng = np.random.default_rng(1)
    x = rng.standard_normal(25); x = x + 1j * rng.standard_normal(25)
    assert np.allclose(dft25(x, -1), np.fft.fft(x)) and np.allclose(dft25(x, +1), np.fft.ifft(x) * 25)
    assert np.allclose(dft20(x[:20], -1), np.fft.fft(x[:20])) and np.allclose(dft20(x[:20], +1), np.fft.ifft(x[:20]) * 20)
    z = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    Zb = forward(z)
    Z = np.fft.fft(z)
    ref = np.array([[Z[kl + P * k3] for k3 in range(R3)] for kl in range(P)])
    assert np.allclose(Zb, ref), 'forward'
    assert np.allclose(inverse(Zb), z * M), 'inverse'
    # full pipeline against the definition: A(n) = sum_k wf_k V_k e^{+2 pi i k n / N} (two-sided, Hermitian)
    xr = rng.standard_normal(N)
    K = M + 1
    wf = rng.standard_normal(K) + 1j * rng.standard_normal(K)
    wf[0] = wf[0].real; wf[M] = wf[M].real
    g = rng.random(K)
    V = np.fft.rfft(xr)
    want = np.fft.irfft(wf * V, N) * N
    wt = np.full(K, 2.0); wt[0] = wt[M] = 1.0
    chi_want = np.sum(wt * g * np.abs(V) ** 2)
    zz = xr[0::2] + 1j * xr[1::2]
    Zp, chi = middle(forward(zz), wf, g)
    a = inverse(Zp)
    got = np.empty(N); got[0::2] = a.real; got[1::2] = a.imag
    print('amp max err', np.max(np.abs(got - want)) / np.max(np.abs(want)), 'chi', chi / chi_want)
    assert np.allclose(got, want) and np.isclose(chi, chi_want)
    print('model ok')
