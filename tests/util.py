"""Shared helpers for the parity tests: build a plan the way the host driver
does, run it through the C ABI, compare with oracle results under the stated
fp64 -> fp32 tolerances."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# ---- tolerances (north_star: within 1e-5 relative of the fp64 path) ----------
# amp      : |d| <= 1e-5 |amp| + 1e-4 ampres       (fp32 FFT rounding: ~5e-7 of the largest
#            amplitude in the trace, so a noise-only event next to nothing is bounded in
#            units of the resolution, not of itself)
# chi2     : |d| <= 1e-5 chi2 + 2e-6 chi2nopulse   (chi2 = chi0 - A^2 norm: a pulse of SNR s in
#            a trace of N samples cancels (s^2 + N) / N digits-worth; the absolute term is the
#            fp32 error of the two terms that cancel)
# lowchi2  : |d| <= 1e-5 lowchi2 + 2e-6 chi2nopulse
# t0       : the BIN must match exactly; the float value to 1e-6 relative.  The one exception, opt-in
#            (`lag_amps=`): a NEAR TIE -- the fp64 amplitudes at the two bins agree to TIE_RTOL = 1e-6
#            in A^2 (chi2 = chi0 - A^2 norm: the bins then differ by < 1e-6 A^2 norm in chi2), which
#            is below the fp32 transform's ~5e-7 amplitude error; the event is then compared with
#            the oracle evaluated at the engine's bin.  Classified case: the general fuzz, seed 77,
#            case 28 (24000 samples, muon template, SNR 117): A(3381) and A(3382) differ by 4.6e-8
#            relative, chi2 by 1.3e-8 chi2_0 (tests/test_oracle_kat.py::test_near_tie...)
# timeres  : 2e-5 relative plus the relative error allowed on amp ; ampres 1e-6 relative
AMP_RTOL, AMP_ATOL_SIGMA = 1e-5, 1e-4
CHI_RTOL, CHI_ATOL_CHI0 = 1e-5, 2e-6
TIE_RTOL = 1e-6


def combine_fp32(ev, chans, weights):
    """The combined trace exactly as the device forms it on load (ofx_plan_set_channels:
    d = fp32(w0 s0), then d = fp32(fma(w_c, s_c, d)) in order; the products of two fp32 numbers are
    exact in fp64, so rounding the fp64 sum once more reproduces the fp32 FMA).  The oracle is fed THIS
    trace -- the channel algebra itself is exact-arithmetic-checked by that construction -- and the
    ordinary tolerances below apply to plans with channel algebra as to any other."""
    ev = np.asarray(ev, dtype=np.float32)
    w = [np.float32(x) for x in weights]
    d = (w[0].astype(np.float64) * ev[:, chans[0]].astype(np.float64)).astype(np.float32)
    for c, wc in zip(chans[1:], w[1:]):
        d = (wc.astype(np.float64) * ev[:, c].astype(np.float64) + d.astype(np.float64)).astype(np.float32)
    return d.astype(np.float64)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


# interpolate=True: the sub-sample offset is a ratio of differences of neighbouring
# amplitudes; fp32 rounding of the amplitudes (~1e-6 of their scale) moves it by < 1e-3 sample for
# pulses above ~30 sigma (the error scales as 1 / SNR: 1.05e-3 sample seen at SNR 28, 1e-4 at 300).
T0_INTERP_ATOL_SAMPLES = 1e-3


def t0_interp_tol_samples(am, a0, ap, amax):
    """Tolerance on the interpolated offset (samples) from the conditioning of the parabola: the vertex
    x = (ap^2 - am^2) / (2 (2 a0^2 - am^2 - ap^2)) moves by ~ 4 eps / c for a relative error eps of the
    amplitudes, c = (2 a0^2 - am^2 - ap^2) / a0^2 being the relative curvature of A^2 at the peak.
    eps = 1e-6 amax / |a0|: the fp32 transform's error is ~5e-7 of the LARGEST amplitude of the trace.
    For a template that matches the pulse c = 2 / (fs sigma_t SNR)^2 (3e-1 for the pulse template at
    1.25 MHz: 1.4e-5 sample at any SNR above the noise floor, 1 / SNR below it); a template that does
    not match the pulse gives a flat top -- the soak case of round 2 (general fuzz seed 101, case 113:
    500 samples, glitch filter on a pulse of SNR 154, amplitudes 154.298 / 154.339 / 154.337 sigma at
    the three bins, c = 5.6e-4): 7e-3 sample allowed, 1.02e-3 seen on the ROCFFT and LDS engines alike;
    rounding the fp64 amplitudes to fp32 alone moves the vertex by 6.5e-5."""
    am, a0, ap = (np.asarray(v, dtype=np.float64) for v in (am, a0, ap))
    c = np.abs(2 * a0 ** 2 - am ** 2 - ap ** 2) / np.maximum(a0 ** 2, 1e-300)
    eps = 1e-6 * np.asarray(amax, dtype=np.float64) / np.maximum(np.abs(a0), 1e-300)
    return 4.0 * eps / np.maximum(c, 1e-300) + 1e-6


def check_search(out, off, ref, prefix, ampres, fs, what="", interpolated=False, lag_amps=None,
                 lowchi2_fcutoff=10000.0):
    """out: [B, row] float64 engine output, ref: dict (golden or oracle).
    lag_amps: optional [B, N] fp64 amplitudes of the oracle at every rolled index: a differing t0 bin
    is then accepted if it is a near tie (TIE_RTOL above), and the event is checked on amp / chi2 at
    the engine's bin; returns the number of such ties."""
    g = lambda k: np.asarray(ref[f"{prefix}{k}"], dtype=np.float64)
    chi0 = g("chi2nopulse")
    idx = g("index").astype(np.int64)
    got_idx = out[:, off + 7].astype(np.int64)
    n_ties = 0
    if lag_amps is not None and not np.array_equal(got_idx, idx):
        bad = np.nonzero(got_idx != idx)[0]
        a_got = lag_amps[bad, got_idx[bad]]
        a_ref = lag_amps[bad, idx[bad]]
        tie = np.abs(a_got ** 2 - a_ref ** 2) <= TIE_RTOL * a_ref ** 2
        assert tie.all(), f"{what}: t0 bins differ at {bad[~tie]} and are not near ties"
        # amplitude at the engine's bin against the oracle's amplitude there; the rest of the row is
        # compared on the other events
        assert np.all(np.abs(out[bad, off + 0] - a_got) <= AMP_RTOL * np.abs(a_got) + AMP_ATOL_SIGMA * ampres), \
            f"{what}: amp at a tied bin"
        keep = np.ones(len(idx), bool)
        keep[bad] = False
        n_ties = len(bad)
        if not keep.any():
            return n_ties
        out = out[keep]
        ref = {k: np.asarray(v)[keep] if np.ndim(v) else v for k, v in ref.items()}
        g = lambda k: np.asarray(ref[f"{prefix}{k}"], dtype=np.float64)
        chi0 = g("chi2nopulse")
        idx = g("index").astype(np.int64)
        got_idx = out[:, off + 7].astype(np.int64)
    assert np.array_equal(got_idx, idx), f"{what}: t0 bins differ at {np.nonzero(got_idx != idx)[0]}"
    amp = g("amp")
    assert np.all(np.abs(out[:, off + 0] - amp) <= AMP_RTOL * np.abs(amp) + AMP_ATOL_SIGMA * ampres), \
        f"{what}: amp"
    t0 = g("t0")
    if interpolated and lag_amps is not None:
        la = lag_amps[keep] if n_ties else lag_amps
        N = la.shape[1]
        r = np.arange(len(idx))
        inner = (idx > 0) & (idx < N - 1)
        tol = np.full(len(idx), T0_INTERP_ATOL_SAMPLES)
        tol[inner] = t0_interp_tol_samples(la[r[inner], idx[inner] - 1], la[r[inner], idx[inner]],
                                           la[r[inner], idx[inner] + 1], np.abs(la[inner]).max(axis=1))
        assert np.all(np.abs(out[:, off + 1] - t0) * fs <= tol), \
            f"{what}: t0 (worst {np.max(np.abs(out[:, off + 1] - t0) * fs / tol):.2f} of the tolerance)"
        # lowchi2 is evaluated AT the refined t0: |d lowchi2 / d t0| <= 2 sqrt(lowchi2) SNR omega_cut, so
        # an offset allowed to move by `tol` samples moves it by this much
        t0_slack = 2.0 * np.sqrt(np.abs(g("lowchi2"))) * np.abs(amp) / ampres * \
            (2.0 * np.pi * lowchi2_fcutoff / fs) * tol
    elif interpolated:
        assert np.all(np.abs(out[:, off + 1] - t0) <= T0_INTERP_ATOL_SAMPLES / fs), f"{what}: t0"
    else:
        assert np.all(np.abs(out[:, off + 1] - t0) <= 1e-6 * np.abs(t0) + 1e-12), f"{what}: t0"
    chi2 = g("chi2")
    lim = CHI_RTOL * np.abs(chi2) + CHI_ATOL_CHI0 * np.where(np.isnan(chi0), np.abs(chi2) + amp ** 2 / ampres ** 2, chi0)
    assert np.all(np.abs(out[:, off + 2] - chi2) <= lim), f"{what}: chi2 {np.max(np.abs(out[:, off + 2] - chi2) / lim)}"
    low = g("lowchi2")
    if not (interpolated and lag_amps is not None):
        t0_slack = 0.0
    assert np.all(np.abs(out[:, off + 3] - low) <= CHI_RTOL * np.abs(low) + lim + t0_slack), f"{what}: lowchi2"
    if not np.all(np.isnan(chi0)):
        assert np.allclose(out[:, off + 4], chi0, rtol=CHI_RTOL), f"{what}: chi2nopulse"
        tr = g("timeres")
        # timeres = 1 / sqrt(amp^2 tres_sum): inherits the relative error allowed on amp
        rel = 2e-5 + (AMP_RTOL * np.abs(amp) + AMP_ATOL_SIGMA * ampres) / np.maximum(np.abs(amp), 1e-300)
        bad = ~(np.abs(out[:, off + 6] - tr) <= rel * np.abs(tr))
        assert not bad.any(), f"{what}: timeres {out[bad, off + 6]} vs {tr[bad]} (amp {amp[bad]})"
    assert np.allclose(out[:, off + 5], ampres, rtol=1e-6), f"{what}: ampres"
    return n_ties


def check_td(out, off, ref, i, traces, what=""):
    scale = float(np.max(np.abs(traces)))
    for j, k in enumerate(("baseline", "integral", "maximum", "minimum")):
        r = np.asarray(ref[f"td{i}_{k}"], dtype=np.float64)
        atol = 2e-6 * scale
        if k == "integral":
            w = ref["td_windows"][i]
            atol = 2e-6 * scale * max(1, (w[1] - w[0])) / float(ref["fs"])
        if k in ("maximum", "minimum"):
            assert np.array_equal(out[:, off + j], r.astype(np.float32).astype(np.float64)), f"{what}: {k}"
        else:
            assert np.all(np.abs(out[:, off + j] - r) <= 1e-5 * np.abs(r) + atol), f"{what}: {k}"
