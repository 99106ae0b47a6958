#!/usr/bin/env python3
"""Regenerate the of1x1 columns of tests/golden/golden_n*.npz from QETpy itself and diff them against
the oracle's -- the step that turns "parity unpinned" into pinned (last row;
DESIGN.md section 3).  Build container only: it never travels to the GPU box and nothing imports it.

QETpy (spice-herald/QETpy, `qetpy>=1.8.6`, detprocess setup.py:73) is NOT installed in this image and
there is no network, so today this script prints that and exits with status 2.  The day
`import qetpy` works, run

    python tests/golden/make_golden_from_qetpy.py            # diff only
    python tests/golden/make_golden_from_qetpy.py --write    # also write golden_qetpy_n*.npz

It drives QETpy through exactly the call sequence of the reference, nothing else:

  one-time, per fixture      detprocess/process/processing_data.py:278-381
      qp.OFBase(sample_rate, verbose=True)
      .set_csd(chan, csd, coupling='AC', ignored_frequency_peaks=None, ignore_harmonics=...)
      .add_template(chan, template, template_tag=..., pretrigger_samples=..., integralnorm=False,
                    overwrite=True)
      .calc_phi(chan, template_tag)
  per event                  processing_data.py:731-772
      .clear_signal(); .update_signal(chan, trace, calc_fft=True)
      .calc_signal_filt(chan); .calc_signal_filt_td(chan)
  per algorithm              detprocess/core/algorithms.py:331-341, 410-421, 533-558
      qp.OF1x1(of_base=, channel=, template_tag=).calc(...)
      .get_result_nodelay() / .get_result_withdelay() / .get_chisq_nopulse() /
      .get_energy_resolution() / .get_time_resolution()

and compares every column the fixtures hold (amp, t0, chi2, lowchi2, chi2nopulse, ampres, timeres; the
rolled bin from t0) with the oracle's under tests/util.py's tolerances tightened to fp64 (1e-9
relative): both sides are fp64 here, a difference is a difference of convention (SURVEY.md Appendix C:
window end points, lowchi2 band edge, interpolation), and the report names the column and event.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CHAN = "chan0"
TAG = "default"
RTOL = 1e-9
FIXTURES = ("golden_n4096.npz", "golden_n4096_pre1000.npz", "golden_n25000.npz", "golden_n32768.npz")


def of_base_for(qp, g):
    """processing_data.py:278-381 for one channel, one template tag, coupling 'AC' (the default,
    processing_data.py:252-254), no ignored frequency peaks."""
    fs = float(g["fs"])
    ofb = qp.OFBase(fs, verbose=False)
    # the filter file stores the two-sided PSD in fftfreq order (filterdata.py:673-676); a single
    # channel's "csd" is that PSD (processing_data.py:296-300: get_csd(chan))
    ofb.set_csd(CHAN, np.asarray(g["psd"], dtype=np.float64), coupling="AC",
                ignored_frequency_peaks=None, ignore_harmonics=False)
    ofb.add_template(CHAN, np.asarray(g["template"], dtype=np.float64), template_tag=TAG,
                     pretrigger_samples=int(g["pre"]), integralnorm=False, overwrite=True)
    if ofb.phi(CHAN, TAG) is None:
        ofb.calc_phi(CHAN, TAG)
    return ofb


def run_event(qp, ofb, trace, g):
    """processing_data.py:731-772, then the three extractors of algorithms.py on this event."""
    ofb.clear_signal()
    ofb.update_signal(CHAN, np.asarray(trace, dtype=np.float64), calc_fft=True)
    ofb.calc_signal_filt(CHAN)
    ofb.calc_signal_filt_td(CHAN)
    out = {}
    # of1x1_nodelay, algorithms.py:331-341
    OF = qp.OF1x1(of_base=ofb, channel=CHAN, template_tag=TAG)
    OF.calc(lgc_fit_withdelay=False, lgc_fit_nodelay=True, lowchi2_fcutoff=10000.0)
    amp, t0, chi2, low = OF.get_result_nodelay()
    out["nodelay"] = dict(amp=amp, t0=0.0, chi2=chi2, lowchi2=low)
    # of1x1_unconstrained, algorithms.py:410-421
    OF = qp.OF1x1(of_base=ofb, channel=CHAN, template_tag=TAG)
    OF.calc(lowchi2_fcutoff=10000.0, interpolate_t0=False, lgc_fit_withdelay=True,
            lgc_fit_nodelay=False, lgc_plot=False)
    amp, t0, chi2, low = OF.get_result_withdelay()
    out["unconstrained"] = dict(amp=amp, t0=t0, chi2=chi2, lowchi2=low)
    # of1x1_constrained, algorithms.py:533-558 (inside and outside the window)
    w = float(g["win_us"])
    lo, hi = int(g["window_lo"]), int(g["window_hi"])
    for name, outside in (("constrained", False), ("outside", True)):
        OF = qp.OF1x1(of_base=ofb, channel=CHAN, template_tag=TAG)
        # features.py:780-785 always injects the indices next to the usec keys
        OF.calc(window_min_from_trig_usec=-w, window_max_from_trig_usec=w,
                window_min_index=lo, window_max_index=hi, lowchi2_fcutoff=10000.0,
                interpolate_t0=False, lgc_outside_window=outside, lgc_fit_withdelay=True,
                lgc_fit_nodelay=False, lgc_plot=False)
        amp, t0, chi2, low = OF.get_result_withdelay()
        out[name] = dict(amp=amp, t0=t0, chi2=chi2, lowchi2=low,
                         chi2nopulse=OF.get_chisq_nopulse(), ampres=OF.get_energy_resolution(),
                         timeres=OF.get_time_resolution())
    return out


def main():
    try:
        import qetpy as qp
    except ImportError as exc:
        print("make_golden_from_qetpy: QETpy is not importable here (%s).\n"
              "  Nothing was regenerated; the golden fixtures still pin the fp64 restatement only\n"
              "  (DESIGN.md section 3, \"parity unpinned\").  Install qetpy>=1.8.6 in the build\n"
              "  container and run this script again." % exc)
        return 2
    write = "--write" in sys.argv
    bad = 0
    for name in FIXTURES:
        path = os.path.join(HERE, name)
        if not os.path.exists(path):
            continue
        g = dict(np.load(path))
        fs, pre = float(g["fs"]), int(g["pre"])
        traces = np.asarray(g["traces"], dtype=np.float64)      # the float32 inputs, widened
        ofb = of_base_for(qp, g)
        cols = {}
        for b in range(traces.shape[0]):
            ev = run_event(qp, ofb, traces[b], g)
            for mode, d in ev.items():
                for k, v in d.items():
                    cols.setdefault(f"{mode}_{k}", []).append(float(v))
        cols = {k: np.asarray(v) for k, v in cols.items()}
        for mode in ("unconstrained", "constrained", "outside"):
            cols[f"{mode}_index"] = np.rint(cols[f"{mode}_t0"] * fs).astype(np.int64) + pre
        print(f"{name}: {traces.shape[0]} events x {traces.shape[1]} samples")
        for k in sorted(cols):
            if k not in g:
                continue
            want, got = np.asarray(g[k], dtype=np.float64), cols[k]
            ok = np.isclose(got, want, rtol=RTOL, atol=0.0) | (np.isnan(got) & np.isnan(want))
            if k.endswith("_index"):
                ok = got == want.astype(np.int64)
            if not ok.all():
                bad += 1
                w = np.nonzero(~ok)[0]
                print(f"   {k:28s} DIFFERS at {len(w)} event(s), first {w[:5]}: qetpy {got[w[:3]]} "
                      f"oracle {want[w[:3]]}")
            else:
                print(f"   {k:28s} equal to {RTOL:g} relative")
        if write:
            out = dict(g)
            out.update({k: v for k, v in cols.items()})
            out["source"] = np.asarray("qetpy " + getattr(qp, "__version__", "?"))
            np.savez_compressed(os.path.join(HERE, name.replace("golden_", "golden_qetpy_")), **out)
    if bad:
        print(f"{bad} column(s) differ between QETpy and oracle/of1x1.py: fix the oracle (and the conventions "
              f"of DESIGN.md section 3), rerun tests/golden/make_golden.py and every parity test.")
        return 1
    print("QETpy and the oracle agree on every fixture: parity is pinned; say so in DESIGN.md section 3.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
