"""
oracle/ofnxm.py -- CPU fp64 restatement of the N-channel x M-template optimal filter
behind ``FeatureExtractors.ofnxm`` (detprocess/core/algorithms.py:141-274).

TEST INFRASTRUCTURE ONLY (same rule as oracle/of1x1.py): only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.

PARITY UNPINNED.  The arithmetic is QETpy's ``OFBase`` (NxM branch) and ``OFnxm``
(``qp.OFnxm(of_base=, channels=, template_tag=).calc()``, ``get_fit_withdelay`` /
``get_fit_nodelay``: algorithms.py:241-262); QETpy is neither in /root/reference nor in
this image, and the reference holds no fixture for the path.  Restated here is the
published joint fit of M amplitudes sharing one delay, in the normalisation-free form of
oracle/of1x1.py, to which it reduces for N = M = 1 (checked in tests/test_ofnxm.py).
In-tree evidence used: templates are ``[n_channels, m_amplitudes, samples]`` and the CSD
``[n_channels, n_channels, frequencies]``, two-sided, A^2/Hz (oftrigger.py:375-386,
:407-445; processing_data.py:294-326, algorithms.py:209 ``template.shape[1]`` = number of
amplitudes); the weight matrix and its inverse ("iweight", whose diagonal gives the
amplitude resolutions) oftrigger.py:481-499; DC unused under AC coupling (:486-488).

Conventions
-----------
v_b[n] traces of the N channels, s_bm[n] templates, V = FFT(v), S = FFT(s) (NumPy);
C_k the N x N two-sided CSD at fftfreq bin k, Ci_k = C_k^-1 (0 at DC for AC coupling and
at notched bins).

    phi_mb(k)  = sum_a conj(S_am(k)) Ci_ab(k)
    P_mm'      = Re sum_k sum_b phi_mb(k) S_bm'(k) / (N fs)          ("weight")
    q_m(n)     = sum_k e^{+2 pi i k n / N} sum_b phi_mb(k) V_b(k) / (N fs)
    amps(n)    = P^-1 q(n)
    chi2_0     = Re sum_k V^H Ci V / (N fs)
    chi2(n)    = chi2_0 - q(n)^T P^-1 q(n)
    rolled index i = (n + pretrigger) mod N ;  t0 = (i - pretrigger) / fs

no-delay fit: n = 0.  Delay fit: first minimum of chi2 in rolled order over the window
(same window rule as oracle/of1x1.py ``search_range``; ``lgc_outside_window`` searches
the complement).
"""

import numpy as np

from . import of1x1 as _o1


class NxMFilter:
    """One-time precompute (processing_data.py:294-381 for an ``a|b`` channel)."""

    def __init__(self, templates, csd, fs, pretrigger_samples, coupling="AC",
                 ignored_frequency_peaks=None, ignore_harmonics=False):
        s = np.asarray(templates, dtype=np.float64)
        c = np.asarray(csd, dtype=np.complex128)
        if s.ndim != 3 or c.ndim != 3 or c.shape[0] != c.shape[1] or c.shape[0] != s.shape[0] \
                or c.shape[2] != s.shape[2]:
            raise ValueError("ERROR: templates must be [N, M, samples] and csd [N, N, samples]")
        self.C, self.M, self.N = s.shape
        self.fs = float(fs)
        self.pre = int(pretrigger_samples)
        N = self.N
        # bins dropped by set_csd (AC coupling, notches): reuse the 1x1 rule on a dummy PSD
        keep = np.isfinite(_o1.effective_psd(np.ones(N), fs, coupling, ignored_frequency_peaks,
                                             ignore_harmonics))
        icov = np.zeros((N, self.C, self.C), dtype=np.complex128)
        ck = np.moveaxis(c, 2, 0)
        icov[keep] = np.linalg.inv(ck[keep])
        self.icov = icov                                   # [k, a, b]
        S = np.fft.fft(s, axis=-1)                         # [a, m, k]
        self.S = S
        self.phi = np.einsum("amk,kab->kmb", np.conj(S), icov)          # [k, m, b]
        P = np.einsum("kmb,bnk->mn", self.phi, S).real / (N * self.fs)
        self.P = 0.5 * (P + P.T)
        self.Pinv = np.linalg.inv(self.P)
        self.ampres = np.sqrt(np.diag(self.Pinv))


def signal_products(filt, x):
    """x [C, N] -> q rolled [M, N], chi2_0."""
    V = np.fft.fft(np.asarray(x, dtype=np.float64), axis=-1)            # [b, k]
    Q = np.einsum("kmb,bk->mk", filt.phi, V) / (filt.N * filt.fs)
    q = np.fft.ifft(Q, axis=-1).real * filt.N
    chi0 = float(np.einsum("ak,kab,bk->", np.conj(V), filt.icov, V).real / (filt.N * filt.fs))
    return np.roll(q, filt.pre, axis=-1), chi0


def fit(filt, x, window_min_from_trig_usec=None, window_max_from_trig_usec=None,
        window_min_index=None, window_max_index=None, lgc_outside_window=False,
        window_policy="qetpy", interpolate_t0=False):
    """One event: dict with the no-delay and the windowed delay fit.  interpolate_t0
    (algorithms.py:152, 259 -> qp.OFnxm.get_fit_withdelay, unseen): restated by analogy with
    the single-template helper (oracle/of1x1.py interpolate_of): vertex of the parabola through
    chi2 at the three bins around the discrete minimum, every amplitude from its own parabola at
    that offset; same guards."""
    q, chi0 = signal_products(filt, x)
    red = np.einsum("mn,ml,ln->n", q, filt.Pinv, q)
    chi2 = chi0 - red
    lo, hi = _o1.search_range(filt, window_min_from_trig_usec, window_max_from_trig_usec,
                              window_min_index, window_max_index, window_policy)
    i = _o1._argmin_window(chi2, lo, hi, lgc_outside_window)
    out = {"amps_nodelay": filt.Pinv @ q[:, filt.pre], "chi2_nodelay": chi2[filt.pre]}
    if i is None:
        out.update(amps=np.full(filt.M, -999999.0), t0=-999999.0, chi2=-999999.0, index=-1)
    else:
        amps, t0, c2 = filt.Pinv @ q[:, i], (i - filt.pre) / filt.fs, chi2[i]
        if interpolate_t0 and 0 < i < filt.N - 1:
            y0, y1, y2 = chi2[i - 1], chi2[i], chi2[i + 1]
            den = y0 - 2.0 * y1 + y2
            if den > 0.0 and abs(0.5 * (y0 - y2) / den) <= 1.0:
                xv = 0.5 * (y0 - y2) / den
                am, ap = filt.Pinv @ q[:, i - 1], filt.Pinv @ q[:, i + 1]
                amps = amps + 0.5 * (ap - am) * xv + 0.5 * (am - 2.0 * amps + ap) * xv * xv
                t0 = t0 + xv / filt.fs
                c2 = y1 - 0.125 * (y0 - y2) ** 2 / den
        out.update(amps=amps, t0=t0, chi2=c2, index=i)
    out["chi2_0"] = chi0
    return out


def process_events(filt, events, **kw):
    """events [B, C, N] -> dict of arrays (amps [B, M], t0, chi2, index, amps_nodelay, ...)."""
    rows = [fit(filt, e, **kw) for e in np.asarray(events)]
    return {k: np.array([r[k] for r in rows]) for k in rows[0]}
