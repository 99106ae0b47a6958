"""One-time host-side filter precompute (fp64 NumPy).

Product counterpart of ``qp.OFBase.set_csd / add_template / calc_phi`` as
driven by ``detprocess/process/processing_data.py:294-381``: runs once per
(channel, template_tag, csd_tag, coupling, N, pretrigger), stays on the host
 and produces the one-sided tables that
``ofx_plan_set_filter`` rounds to fp32 device memory.

Definitions (QETpy >= 1.8.6 OFBase restated in
normalisation-free form): S = FFT(template), J two-sided PSD (A^2/Hz, fftfreq
order), J = inf at DC for AC coupling and at notched bins,

    norm     = sum_k |S_k|^2 / J_k / (N fs)
    wf_k     = conj(S_k) / J_k / (N fs) / norm        ->  A(n) = sum_k wf_k V_k e^{+2 pi i k n / N}
    g_k      = 1 / (J_k N fs)                          ->  chi2_0 = sum_k g_k |V_k|^2
    tres_sum = sum_k (2 pi f_k)^2 |S_k|^2 g_k          ->  timeres = 1/sqrt(A^2 tres_sum)
"""

from dataclasses import dataclass

import numpy as np


def apply_coupling_and_notches(psd, fs, coupling="AC", ignored_frequency_peaks=None,
                               ignore_harmonics=False):
    """PSD edits of ``OFBase.set_csd(chan, csd, coupling=, ignored_frequency_peaks=,
    ignore_harmonics=)`` (processing_data.py:252-272, 321-326; notch semantics
    documented at oftrigger.py:387-390: nearest bin at +f and -f set to inf)."""
    J = np.array(psd, dtype=np.float64).copy()
    if J.ndim != 1:
        raise ValueError("ERROR: psd must be a 1-D two-sided array")
    N = J.shape[0]
    freqs = np.fft.fftfreq(N, d=1.0 / fs)
    if coupling == "AC":
        J[0] = np.inf
    elif coupling != "DC":
        raise ValueError('ERROR: "coupling" should be "AC" or "DC"')
    if ignored_frequency_peaks is not None:
        peaks = ignored_frequency_peaks
        if not isinstance(peaks, (list, tuple, np.ndarray)):
            peaks = [peaks]
        for f0 in peaks:
            f0 = abs(float(f0))
            if f0 == 0.0:
                J[0] = np.inf
                continue
            targets = [f0]
            if ignore_harmonics:
                m = 2
                while f0 * m <= fs / 2.0:
                    targets.append(f0 * m)
                    m += 1
            for f in targets:
                J[int(np.argmin(np.abs(freqs - f)))] = np.inf
                J[int(np.argmin(np.abs(freqs + f)))] = np.inf
    return J


@dataclass
class FilterTables:
    n_samples: int
    fs: float
    pretrigger_samples: int
    wf: np.ndarray         # complex128 [K]
    g: np.ndarray          # float64   [K]
    s: np.ndarray          # complex128 [K]  template FFT
    norm: float
    tres_sum: float
    template: np.ndarray   # float64 [N] time-domain template (for reference)

    @property
    def ampres(self):
        return 1.0 / np.sqrt(self.norm)


def build_filter(template, psd, fs, pretrigger_samples, coupling="AC",
                 ignored_frequency_peaks=None, ignore_harmonics=False,
                 integralnorm=False):
    """Precompute the one-sided filter tables for one (template, PSD) pair."""
    template = np.asarray(template, dtype=np.float64)
    if template.ndim != 1:
        raise ValueError("ERROR: template must be a 1-D array")
    N = template.shape[0]
    psd = np.asarray(psd, dtype=np.float64)
    if psd.shape[-1] != N:
        # processing_data.py:312-318 / 351-358
        raise ValueError(f"ERROR: Number of samples is not consistent between "
                         f"template (={N}) and psd (={psd.shape[-1]})!")
    if N % 2:
        raise ValueError("ERROR: odd trace lengths are not supported")
    if np.any(psd[np.isfinite(psd)] < 0):
        raise ValueError("ERROR: psd must be non-negative")
    J = apply_coupling_and_notches(psd, fs, coupling, ignored_frequency_peaks,
                                   ignore_harmonics)
    K = N // 2 + 1
    # two-sided -> one-sided: the PSD of real noise is symmetric; use the
    # non-negative-frequency half (bin N/2 is the Nyquist bin of fftfreq order)
    with np.errstate(divide="ignore"):
        invJ = np.where(np.isinf(J), 0.0, 1.0 / J)
    if not np.allclose(invJ[1:N // 2], invJ[:N // 2:-1], rtol=1e-6, atol=0):
        raise ValueError("ERROR: psd is not symmetric in +/- frequency; a two-sided "
                         "(unfolded) PSD in fftfreq order is required "
                         "(filterdata.py:673-676)")
    S = np.fft.rfft(template)
    if integralnorm:
        S = S / S[0]
    w = np.full(K, 2.0)
    w[0] = 1.0
    w[-1] = 1.0
    nfs = N * float(fs)
    g = invJ[:K] / nfs
    norm = float(np.sum(w * np.abs(S) ** 2 * g))
    if not norm > 0:
        raise ValueError("ERROR: filter norm is not positive (empty template or "
                         "infinite PSD everywhere)")
    wf = np.conj(S) * g / norm
    f = np.arange(K) * (fs / N)
    tres = float(np.sum(w * (2 * np.pi * f) ** 2 * np.abs(S) ** 2 * g))
    return FilterTables(N, float(fs), int(pretrigger_samples), wf, g, S, norm, tres,
                        template)
