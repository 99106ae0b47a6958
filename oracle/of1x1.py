"""
oracle/of1x1.py -- CPU fp64 restatement of the detprocess + QETpy of1x1 hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.  The product
(``detprocess_amd``) never imports it and has no CPU fallback.

PARITY UNPINNED.  The arithmetic of this path lives in QETpy
(``spice-herald/QETpy``, constraint ``qetpy>=1.8.6`` -- reference ``setup.py:73``,
no exact pin, no lock file), which is neither vendored in ``/root/reference``
nor installed in this image.  The reference holds no tests, golden vectors or
fixtures for the path.  What pins this file instead:
analytic known-answer tests (``tests/test_oracle_kat.py``) and the self-made
fixtures under ``tests/golden`` (made by ``tests/golden/make_golden.py`` from
this file).  All GPU parity is "vs. our fp64 restatement of the QETpy path".

What is restated, and from where
--------------------------------
* The call sequence, kwargs, output keys and sentinels of
  ``detprocess/core/algorithms.py:277-350`` (of1x1_nodelay), ``:354-432``
  (of1x1_unconstrained), ``:435-570`` (of1x1_constrained), ``:650-885``
  (baseline / integral / maximum / minimum, end-exclusive slices).
* The one-time precompute order of
  ``detprocess/process/processing_data.py:155-433`` (set_csd with AC coupling
  default ``:252-254``, notches ``:258-272``, add_template ``:369-376``,
  calc_phi ``:379-381``) and the per-event order of ``:712-772``
  (clear_signal, update_signal(calc_fft=True), calc_signal_filt,
  calc_signal_filt_td).
* The window-index helper ``detprocess/process/features.py:1243-1344``.
* QETpy's published single-template optimal filter (OFBase / OF1x1), restated
  in normalisation-free physical form.  In-tree evidence
  for the conventions: ``algorithms.py:1013`` (signal_fft is FFT/N),
  ``oftrigger.py:387-390`` (notch = nearest +/- bin -> inf), ``oftrigger.py:492``
  (DC of phi is not used), ``noise.py:344-346`` + ``filterdata.py:673-676``
  (PSD two-sided, unfolded, A^2/Hz, fftfreq order).

Conventions (each an explicit choice; see DESIGN.md "Parity risks")
-------------------------------------------------------------------
v[n] trace (A), s[n] template, N samples, fs; V = FFT(v), S = FFT(s) (NumPy,
unnormalised); J[k] two-sided PSD in fftfreq order with J = inf at DC for AC
coupling and at notched bins; df = fs/N.

    norm      = sum_k |S_k|^2 / J_k / (N fs)
    A(n)      = sum_k conj(S_k) V_k e^{+2 pi i k n / N} / J_k / (N fs) / norm
    chi2_0    = sum_k |V_k|^2 / J_k / (N fs)
    chi2(n)   = chi2_0 - A(n)^2 norm
    rolled    : index i = (n + pretrigger) mod N ;  t0 = (i - pretrigger)/fs
    lowchi2   = sum_{|f_k| <= fcut} |V_k - A e^{-2 pi i f_k t0} S_k|^2 / J_k / (N fs)
    ampres    = 1/sqrt(norm)
    timeres   = 1/sqrt(A^2 sum_k (2 pi f_k)^2 |S_k|^2 / J_k / (N fs))

arg-min: first minimum in rolled order (NumPy ``argmin``).
Window (``window_policy``):
  'qetpy'  (default) -- restated QETpy >= 1.8 ``OFBase.get_fit_withdelay``:
           ``window_*_from_trig_usec`` take precedence over the indices,
           min = floor(pre + us fs 1e-6), max = ceil(pre + us fs 1e-6),
           lags searched = [min, max) (half-open), clipped to [0, N).
  'index'  -- SURVEY.md Appendix C-2 alternative: the indices computed by
           ``features.py:1243-1344``, inclusive on both ends.
"""

from math import ceil, floor

import numpy as np

SENTINEL = -999999.0  # algorithms.py:319-327, 398-407, 517-529, 683-688


# ----------------------------------------------------------------------------
# one-time precompute  (processing_data.py:155-433 -> OFBase.set_csd /
# add_template / calc_phi)
# ----------------------------------------------------------------------------

def effective_psd(psd, fs, coupling="AC", ignored_frequency_peaks=None,
                  ignore_harmonics=False):
    """Two-sided PSD with the OFBase.set_csd edits applied.

    coupling 'AC' -> J[0] = inf (processing_data.py:252-254 default);
    ignored_frequency_peaks: nearest bin at +f and -f -> inf
    (oftrigger.py:387-390); ignore_harmonics: every multiple up to Nyquist.
    """
    J = np.array(psd, dtype=np.float64).copy()
    N = J.shape[-1]
    freqs = np.fft.fftfreq(N, d=1.0 / fs)
    if coupling == "AC":
        J[0] = np.inf
    elif coupling != "DC":
        raise ValueError('ERROR: coupling must be "AC" or "DC"')
    if ignored_frequency_peaks is not None:
        peaks = ignored_frequency_peaks
        if not isinstance(peaks, (list, tuple, np.ndarray)):
            peaks = [peaks]
        for f0 in peaks:
            f0 = abs(float(f0))
            if f0 == 0.0:
                J[0] = np.inf
                continue
            fl = [f0]
            if ignore_harmonics:
                m = 2
                while f0 * m <= fs / 2.0:
                    fl.append(f0 * m)
                    m += 1
            for f in fl:
                J[int(np.argmin(np.abs(freqs - f)))] = np.inf
                J[int(np.argmin(np.abs(freqs + f)))] = np.inf
    return J


class OFFilter:
    """Precomputed optimal filter for one (channel, template_tag, csd_tag)."""

    def __init__(self, template, psd, fs, pretrigger_samples, coupling="AC",
                 ignored_frequency_peaks=None, ignore_harmonics=False,
                 integralnorm=False):
        template = np.asarray(template, dtype=np.float64)
        if template.ndim != 1:
            raise ValueError("ERROR: template must be 1-D")
        if np.asarray(psd).shape[-1] != template.shape[-1]:
            # processing_data.py:312-318, 351-358
            raise ValueError("ERROR: Number of samples is not consistent "
                             "between template and psd")
        self.N = int(template.shape[-1])
        self.fs = float(fs)
        self.pre = int(pretrigger_samples)
        self.freqs = np.fft.fftfreq(self.N, d=1.0 / self.fs)
        self.J = effective_psd(psd, fs, coupling, ignored_frequency_peaks,
                               ignore_harmonics)
        S = np.fft.fft(template)
        if integralnorm:
            S = S / S[0]
        self.S = S
        self.template = template
        nfs = self.N * self.fs
        self.invJ = np.where(np.isinf(self.J), 0.0, 1.0 / self.J)
        self.norm = float(np.sum(np.abs(S) ** 2 * self.invJ) / nfs)
        # phi / norm in "physical" form:  A(n) = IFFT_unnorm(Wf * V)[n]
        self.Wf = np.conj(S) * self.invJ / nfs / self.norm
        self.g = self.invJ / nfs                       # chi2 weights
        self.ampres = 1.0 / np.sqrt(self.norm)
        self._tres_sum = float(np.sum((2 * np.pi * self.freqs) ** 2
                                      * np.abs(S) ** 2 * self.invJ) / nfs)

    def timeres(self, amp):
        with np.errstate(divide="ignore", invalid="ignore"):
            return 1.0 / np.sqrt(amp ** 2 * self._tres_sum)


# ----------------------------------------------------------------------------
# per-event arithmetic (processing_data.py:712-772 then algorithms.py)
# ----------------------------------------------------------------------------

def signal_products(filt, trace):
    """update_signal(calc_fft=True) + calc_signal_filt + calc_signal_filt_td.

    Returns V (unnormalised FFT), chi2_0, and the rolled amplitude / chi2
    arrays (index i = lag + pretrigger).
    """
    v = np.asarray(trace, dtype=np.float64)
    V = np.fft.fft(v)
    amps = np.real(np.fft.ifft(filt.Wf * V)) * filt.N     # A(n), n = lag
    chi0 = float(np.sum(np.abs(V) ** 2 * filt.g))
    chi2 = chi0 - amps ** 2 * filt.norm
    amps_r = np.roll(amps, filt.pre)
    chi2_r = np.roll(chi2, filt.pre)
    return V, chi0, amps_r, chi2_r


def chi2_lowfreq(filt, V, amp, t0, fcutoff):
    """OFBase.get_chisq_lowfreq restated: |f| <= fcutoff, DC weight is 0 (AC)."""
    sel = np.abs(filt.freqs) <= fcutoff
    f = filt.freqs[sel]
    r = V[sel] - amp * np.exp(-2.0j * np.pi * f * t0) * filt.S[sel]
    return float(np.sum(np.abs(r) ** 2 * filt.g[sel]))


def search_range(filt, window_min_from_trig_usec=None,
                 window_max_from_trig_usec=None, window_min_index=None,
                 window_max_index=None, window_policy="qetpy"):
    """Half-open rolled-index range [lo, hi) searched by the delay fit."""
    N, pre, fs = filt.N, filt.pre, filt.fs
    if window_policy == "qetpy":
        lo = None
        if window_min_from_trig_usec is not None:
            lo = floor(pre + window_min_from_trig_usec * fs * 1e-6)
        elif window_min_index is not None:
            lo = int(window_min_index)
        hi = None
        if window_max_from_trig_usec is not None:
            hi = ceil(pre + window_max_from_trig_usec * fs * 1e-6)
        elif window_max_index is not None:
            hi = int(window_max_index)
    elif window_policy == "index":
        lo = None if window_min_index is None else int(window_min_index)
        hi = None if window_max_index is None else int(window_max_index) + 1
    else:
        raise ValueError("ERROR: unknown window_policy")
    if lo is None or lo < 0:
        lo = 0
    if hi is None or hi > N:
        hi = N
    return int(lo), int(hi)


def _argmin_window(chi2_r, lo, hi, outside, mask=None):
    N = chi2_r.shape[0]
    if outside:
        inds = np.concatenate((np.arange(0, lo), np.arange(hi, N)))
    else:
        inds = np.arange(lo, hi)
    if mask is not None:
        inds = inds[mask[inds]]
    if inds.size == 0:
        return None
    return int(inds[int(np.argmin(chi2_r[inds]))])


def interpolate_of(amps_r, chi2_r, ind, dt):
    """3-point parabolic refinement around the discrete minimum (interpolate=True).

    Unpinned (QETpy's helper unseen): vertex of the parabola through the three
    chi2 points; amplitude from the parabola through the three amplitudes at
    the same offset.  No refinement at the array ends, when the three points are
    not convex, or when the vertex lies more than one bin away (the discrete
    minimum of a window can sit on its edge with the true minimum outside: the
    parabola then extrapolates, by hundreds of bins in noise, to meaningless
    amplitudes).
    """
    N = chi2_r.shape[0]
    if ind <= 0 or ind >= N - 1:
        return amps_r[ind], 0.0, chi2_r[ind]
    y0, y1, y2 = chi2_r[ind - 1], chi2_r[ind], chi2_r[ind + 1]
    den = y0 - 2.0 * y1 + y2
    if den <= 0.0:
        return amps_r[ind], 0.0, chi2_r[ind]
    x = 0.5 * (y0 - y2) / den
    if abs(x) > 1.0:
        return amps_r[ind], 0.0, chi2_r[ind]
    chi2 = y1 - 0.125 * (y0 - y2) ** 2 / den
    a0, a1, a2 = amps_r[ind - 1], amps_r[ind], amps_r[ind + 1]
    amp = a1 + 0.5 * (a2 - a0) * x + 0.5 * (a0 - 2.0 * a1 + a2) * x * x
    return amp, x * dt, chi2


def of1x1_nodelay(filt, trace, lowchi2_fcutoff=10000.0):
    """algorithms.py:277-350 -> amp, chi2, lowchi2 at zero delay."""
    V, chi0, amps_r, chi2_r = signal_products(filt, trace)
    amp = float(amps_r[filt.pre])
    chi2 = float(chi2_r[filt.pre])
    low = chi2_lowfreq(filt, V, amp, 0.0, lowchi2_fcutoff)
    return {"amp": amp, "chi2": chi2, "lowchi2": low}


def of1x1_withdelay(filt, trace, window_min_from_trig_usec=None,
                    window_max_from_trig_usec=None, window_min_index=None,
                    window_max_index=None, lgc_outside_window=False,
                    interpolate=False, lowchi2_fcutoff=10000.0,
                    pulse_direction_constraint=0, window_policy="qetpy"):
    """algorithms.py:354-432 (no window) and :435-570 (window) ->
    amp, t0, chi2, lowchi2, chi2nopulse, ampres, timeres (+ the bin index)."""
    V, chi0, amps_r, chi2_r = signal_products(filt, trace)
    lo, hi = search_range(filt, window_min_from_trig_usec,
                          window_max_from_trig_usec, window_min_index,
                          window_max_index, window_policy)
    mask = None
    if pulse_direction_constraint in (1, -1):
        mask = amps_r * pulse_direction_constraint > 0
    ind = _argmin_window(chi2_r, lo, hi, lgc_outside_window, mask)
    if ind is None:
        nan = float("nan")
        return {"amp": nan, "t0": nan, "chi2": nan, "lowchi2": nan,
                "chi2nopulse": chi0, "ampres": filt.ampres, "timeres": nan,
                "index": -1}
    if interpolate:
        amp, dt, chi2 = interpolate_of(amps_r, chi2_r, ind, 1.0 / filt.fs)
        t0 = (ind - filt.pre) / filt.fs + dt
    else:
        amp, chi2 = amps_r[ind], chi2_r[ind]
        t0 = (ind - filt.pre) / filt.fs
    low = chi2_lowfreq(filt, V, amp, t0, lowchi2_fcutoff)
    return {"amp": float(amp), "t0": float(t0), "chi2": float(chi2),
            "lowchi2": low, "chi2nopulse": chi0, "ampres": filt.ampres,
            "timeres": float(filt.timeres(amp)), "index": ind}


# ----------------------------------------------------------------------------
# time-domain features: exact restatement of algorithms.py:650-885
# ----------------------------------------------------------------------------

def _slice(trace, window_min_index, window_max_index):
    trace = np.asarray(trace, dtype=np.float64)
    if window_min_index is None:
        window_min_index = 0
    if window_max_index is None:
        window_max_index = trace.shape[-1] - 1        # algorithms.py:694-695
    return trace[..., window_min_index:window_max_index]   # end-exclusive


def baseline(trace, window_min_index=None, window_max_index=None):
    return np.mean(_slice(trace, window_min_index, window_max_index), axis=-1)


def integral(trace, fs, window_min_index=None, window_max_index=None):
    # np.trapz(x)/fs with unit spacing (algorithms.py:759) == sum - (first+last)/2
    x = _slice(trace, window_min_index, window_max_index)
    return (np.sum(x, axis=-1) - 0.5 * (x[..., 0] + x[..., -1])) / fs


def maximum(trace, window_min_index=None, window_max_index=None):
    return np.amax(_slice(trace, window_min_index, window_max_index), axis=-1)


def minimum(trace, window_min_index=None, window_max_index=None):
    return np.amin(_slice(trace, window_min_index, window_max_index), axis=-1)


def energyabsorbed(trace, fs, vb, i0, rl, window_min_index, window_max_index):
    """algorithms.py:938-943."""
    trace = np.asarray(trace, dtype=np.float64)
    base = trace[..., :window_min_index].mean(axis=-1, keepdims=True)
    it = trace[..., window_min_index:window_max_index] - base
    p0 = it * (vb - 2 * i0 * rl) - it ** 2 * rl
    return (np.sum(p0, axis=-1) - 0.5 * (p0[..., 0] + p0[..., -1])) / fs


def cleanup_freq_ranges(f_lims):
    """utils/utils.py:437-470: normalised [f_low, f_high] list + range names."""
    if not isinstance(f_lims, list):
        f_lims = [f_lims]
    ranges, names = [], []
    for fr in f_lims:
        if isinstance(fr, (int, float)):
            fr = [fr]
        f_low = abs(fr[0])
        if len(fr) == 2:
            f_high = abs(fr[1])
            if f_low > f_high:
                f_low, f_high = f_high, f_low
            name = f"{round(f_low)}_{round(f_high)}"
            if name not in names:
                ranges.append([f_low, f_high])
                names.append(name)
        else:
            name = f"{round(f_low)}"
            if name not in names:
                ranges.append([f_low])
                names.append(name)
    return ranges, names


def get_ind_freq_ranges(freq_ranges, freqs):
    """utils/utils.py:475-504 (indices into the DC-dropped folded arrays)."""
    out = []
    for fr in freq_ranges:
        lo = int(np.argmin(np.abs(freqs - abs(fr[0]))))
        hi = lo + 1
        if len(fr) == 2:
            hi = int(np.argmin(np.abs(freqs - abs(fr[1]))))
        if lo > hi:
            lo, hi = hi, lo
        if lo == hi:
            if hi < len(freqs) - 1:
                hi += 1
            elif lo > 0:
                lo -= 1
            else:
                raise ValueError("Frequency range too narrow or outside bounds.")
        out.append([lo, hi])
    return out


def psd_amp(trace, fs, f_lims):
    """algorithms.py:1001-1044: psd = |FFT/N|^2 N/fs, folded (x2 except DC and
    Nyquist), DC dropped, sqrt, averaged over [ind_low, ind_high)."""
    x = np.asarray(trace, dtype=np.float64)
    n = x.shape[-1]
    V = np.fft.fft(x, axis=-1) / n
    psd = np.abs(V) ** 2 * n / fs
    k = n // 2 + 1
    fold = psd[..., :k].copy()
    if n % 2:
        fold[..., 1:] *= 2.0
    else:
        fold[..., 1:-1] *= 2.0
    freqs = np.fft.rfftfreq(n, d=1.0 / fs)
    amp = np.sqrt(fold[..., 1:])
    ff = freqs[1:]
    ranges, names = cleanup_freq_ranges(f_lims)
    res = {}
    for (lo, hi), name in zip(get_ind_freq_ranges(ranges, ff), names):
        res[name] = np.average(amp[..., lo:hi], axis=-1)
    return res


# ----------------------------------------------------------------------------
# window helper: features.py:1243-1344 (twin at utils/utils.py:189-301)
# ----------------------------------------------------------------------------

def get_window_indices(nb_samples, nb_pretrigger_samples, fs,
                       window_min_from_start_usec=None,
                       window_min_to_end_usec=None,
                       window_min_from_trig_usec=None,
                       window_max_from_start_usec=None,
                       window_max_to_end_usec=None,
                       window_max_from_trig_usec=None, **kwargs):
    min_index = 0
    if window_min_from_start_usec is not None:
        min_index = int(window_min_from_start_usec * fs * 1e-6)
    elif window_min_to_end_usec is not None:
        min_index = nb_samples - abs(int(window_min_to_end_usec * fs * 1e-6)) - 1
    elif window_min_from_trig_usec is not None:
        min_index = nb_pretrigger_samples + int(window_min_from_trig_usec * fs * 1e-6)
    min_index = min(max(min_index, 0), nb_samples - 1)

    max_index = nb_samples - 1
    if window_max_from_start_usec is not None:
        max_index = int(window_max_from_start_usec * fs * 1e-6)
    elif window_max_to_end_usec is not None:
        max_index = nb_samples - abs(int(window_max_to_end_usec * fs * 1e-6)) - 1
    elif window_max_from_trig_usec is not None:
        max_index = nb_pretrigger_samples + int(window_max_from_trig_usec * fs * 1e-6)
    max_index = min(max(max_index, 0), nb_samples - 1)

    if max_index < min_index:
        raise ValueError("ERROR window calculation: max index smaller than min!"
                         "Check configuration!")
    return min_index, max_index


# ----------------------------------------------------------------------------
# batch driver used by tests and by bench.py's cpu_baseline leg: the per-event
# loop the reference runs (features.py:533-851), one FFT pair per trace.
# ----------------------------------------------------------------------------

OF_COLUMNS = ("amp", "t0", "chi2", "lowchi2", "chi2nopulse", "ampres", "timeres")


def process_events(filt, traces, mode="unconstrained", lowchi2_fcutoff=10000.0,
                   interpolate=False, **window_kwargs):
    """Per-event loop; returns dict of float64 arrays [B] (+ 'index')."""
    traces = np.asarray(traces)
    B = traces.shape[0]
    out = {k: np.empty(B) for k in OF_COLUMNS}
    out["index"] = np.empty(B, dtype=np.int64)
    for b in range(B):
        if mode == "nodelay":
            r = of1x1_nodelay(filt, traces[b], lowchi2_fcutoff)
            r.update(t0=0.0, chi2nopulse=np.nan, ampres=filt.ampres,
                     timeres=np.nan, index=filt.pre)
        elif mode == "unconstrained":
            r = of1x1_withdelay(filt, traces[b], lowchi2_fcutoff=lowchi2_fcutoff,
                                interpolate=interpolate)
        elif mode == "constrained":
            r = of1x1_withdelay(filt, traces[b], lowchi2_fcutoff=lowchi2_fcutoff,
                                interpolate=interpolate, **window_kwargs)
        else:
            raise ValueError("unknown mode")
        for k in OF_COLUMNS:
            out[k][b] = r[k]
        out["index"][b] = r["index"]
    return out
