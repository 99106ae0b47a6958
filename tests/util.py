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
# t0       : the BIN must match exactly; the float value to 1e-6 relative
# timeres  : 2e-5 relative plus the relative error allowed on amp ; ampres 1e-6 relative
AMP_RTOL, AMP_ATOL_SIGMA = 1e-5, 1e-4
CHI_RTOL, CHI_ATOL_CHI0 = 1e-5, 2e-6


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


# interpolate=True: the sub-sample offset is a ratio of differences of neighbouring
# amplitudes; fp32 rounding of the amplitudes (~1e-6 of their scale) moves it by < 1e-3 sample for
# pulses above ~30 sigma (the error scales as 1 / SNR: 1.05e-3 sample seen at SNR 28, 1e-4 at 300).
T0_INTERP_ATOL_SAMPLES = 1e-3


def check_search(out, off, ref, prefix, ampres, fs, what="", interpolated=False):
    """out: [B, row] float64 engine output, ref: dict (golden or oracle)."""
    g = lambda k: np.asarray(ref[f"{prefix}{k}"], dtype=np.float64)
    chi0 = g("chi2nopulse")
    idx = g("index").astype(np.int64)
    got_idx = out[:, off + 7].astype(np.int64)
    assert np.array_equal(got_idx, idx), f"{what}: t0 bins differ at {np.nonzero(got_idx != idx)[0]}"
    amp = g("amp")
    assert np.all(np.abs(out[:, off + 0] - amp) <= AMP_RTOL * np.abs(amp) + AMP_ATOL_SIGMA * ampres), \
        f"{what}: amp"
    t0 = g("t0")
    if interpolated:
        assert np.all(np.abs(out[:, off + 1] - t0) <= T0_INTERP_ATOL_SAMPLES / fs), f"{what}: t0"
    else:
        assert np.all(np.abs(out[:, off + 1] - t0) <= 1e-6 * np.abs(t0) + 1e-12), f"{what}: t0"
    chi2 = g("chi2")
    lim = CHI_RTOL * np.abs(chi2) + CHI_ATOL_CHI0 * np.where(np.isnan(chi0), np.abs(chi2) + amp ** 2 / ampres ** 2, chi0)
    assert np.all(np.abs(out[:, off + 2] - chi2) <= lim), f"{what}: chi2 {np.max(np.abs(out[:, off + 2] - chi2) / lim)}"
    low = g("lowchi2")
    assert np.all(np.abs(out[:, off + 3] - low) <= CHI_RTOL * np.abs(low) + lim), f"{what}: lowchi2"
    if not np.all(np.isnan(chi0)):
        assert np.allclose(out[:, off + 4], chi0, rtol=CHI_RTOL), f"{what}: chi2nopulse"
        tr = g("timeres")
        # timeres = 1 / sqrt(amp^2 tres_sum): inherits the relative error allowed on amp
        rel = 2e-5 + (AMP_RTOL * np.abs(amp) + AMP_ATOL_SIGMA * ampres) / np.maximum(np.abs(amp), 1e-300)
        bad = ~(np.abs(out[:, off + 6] - tr) <= rel * np.abs(tr))
        assert not bad.any(), f"{what}: timeres {out[bad, off + 6]} vs {tr[bad]} (amp {amp[bad]})"
    assert np.allclose(out[:, off + 5], ampres, rtol=1e-6), f"{what}: ampres"


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
