"""Synthetic template / PSD / trace recipes.

Host-side NumPy; used by bench.py, the tests and the golden-fixture script.
Nothing here is on the hot path.
"""

import numpy as np

# reference example shape: examples/trigger/optimal_filter_trigger.ipynb cell 9
_AMPS = (1.67e-15, 1.32e-15, 2.39e-17)
_FALL = (44.6e-6, 147.5e-6, 3872.9e-6)
_RISE = 8.79e-6
FS_DEFAULT = 1.25e6


def two_pole(t, amp, rise, fall):
    out = np.zeros_like(t)
    m = t >= 0
    out[m] = amp * (np.exp(-t[m] / fall) - np.exp(-t[m] / rise))
    return out


def make_template(n_samples, n_pretrigger, fs=FS_DEFAULT, kind="pulse"):
    """Sum-of-two-poles template, max-normalised to 1, onset at n_pretrigger."""
    t = (np.arange(n_samples) - n_pretrigger) / fs
    if kind == "pulse":
        s = sum(two_pole(t, a, _RISE, f) for a, f in zip(_AMPS, _FALL))
    elif kind == "glitch":
        s = two_pole(t, 1.0, 2e-6, 20e-6)
    elif kind == "muon":
        s = two_pole(t, 1.0, 50e-6, 5e-3)
    else:
        raise ValueError(kind)
    return s / np.max(np.abs(s))


def make_psd(n_samples, fs=FS_DEFAULT, j0=1e-22, f_c=1e3, f_r=1e5, line_hz=60.0,
             line_factor=30.0):
    """Two-sided PSD (A^2/Hz) in fftfreq order: 1/f knee, roll-off, 60 Hz line."""
    f = np.abs(np.fft.fftfreq(n_samples, d=1.0 / fs))
    fz = np.where(f == 0, fs / n_samples, f)
    J = j0 * (1.0 + f_c / fz) / (1.0 + (f / f_r) ** 2)
    if line_hz:
        k = int(round(line_hz * n_samples / fs))
        if 0 < k < n_samples // 2:
            J[k] *= line_factor
            J[-k] *= line_factor
    return J


def coloured_noise(rng, n_traces, psd, fs):
    """Gaussian noise consistent with the two-sided PSD J: irfft(sqrt(J N fs / 2) xi)."""
    N = psd.shape[0]
    K = N // 2 + 1
    J1 = psd[:K].copy()
    scale = np.sqrt(J1 * N * fs / 2.0)
    xi = rng.standard_normal((n_traces, K)) + 1j * rng.standard_normal((n_traces, K))
    spec = scale * xi
    spec[:, 0] = spec[:, 0].real * np.sqrt(2.0)
    spec[:, -1] = spec[:, -1].real * np.sqrt(2.0)
    return np.fft.irfft(spec, n=N, axis=-1)


def make_traces(n_traces, template, psd, fs, ampres, seed=0, pulse_fraction=0.5,
                snr_lo=3.0, snr_hi=300.0, max_delay=2000, coloured=True):
    """v = A roll(template, d) + noise ; returns (traces fp64, amps, delays)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    N = template.shape[0]
    if coloured:
        noise = coloured_noise(rng, n_traces, psd, fs)
    else:
        sigma = np.sqrt(np.median(psd) * fs)
        noise = sigma * rng.standard_normal((n_traces, N))
    has = rng.random(n_traces) < pulse_fraction
    amps = np.where(has, ampres * np.exp(rng.uniform(np.log(snr_lo), np.log(snr_hi),
                                                     n_traces)), 0.0)
    md = min(max_delay, N // 4)
    delays = rng.integers(-md, md + 1, n_traces)
    traces = noise
    for b in range(n_traces):
        if amps[b] != 0.0:
            traces[b] += amps[b] * np.roll(template, delays[b])
    return traces, amps, delays
