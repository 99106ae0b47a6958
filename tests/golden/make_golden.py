#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the fp64 oracle.

The reference itself cannot be imported here (QETpy / vaex / pytesio absent --
SURVEY.md section 8c), so these vectors pin "our fp64 restatement", not QETpy.
Inputs are stored as float32 (exactly what the GPU engine consumes); expected
values are the oracle's fp64 results on those float32 inputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from detprocess_amd import synth          # noqa: E402
from oracle import of1x1 as orc           # noqa: E402

FS = 1.25e6


def make(n_samples, n_traces, seed, pre=None, fname=None):
    pre = n_samples // 2 if pre is None else pre
    tmpl = synth.make_template(n_samples, pre, FS)
    psd = synth.make_psd(n_samples, FS)
    filt = orc.OFFilter(tmpl, psd, FS, pre)
    traces, amps, delays = synth.make_traces(n_traces, tmpl, psd, FS, filt.ampres,
                                             seed=seed, max_delay=n_samples // 8)
    tr32 = traces.astype(np.float32)
    x = tr32.astype(np.float64)
    half = 500 * n_samples // 32768
    win_us = half / FS * 1e6
    out = {"template": tmpl, "psd": psd, "fs": FS, "pre": pre, "traces": tr32,
           "true_amp": amps, "true_delay": delays, "win_us": win_us}
    for mode, kw in (("nodelay", {}), ("unconstrained", {}),
                     ("constrained", dict(window_min_from_trig_usec=-win_us,
                                          window_max_from_trig_usec=win_us)),
                     ("outside", dict(window_min_from_trig_usec=-win_us,
                                      window_max_from_trig_usec=win_us,
                                      lgc_outside_window=True))):
        m = "constrained" if mode == "outside" else mode
        r = orc.process_events(filt, x, m, **kw)
        for k, v in r.items():
            out[f"{mode}_{k}"] = v
    lo, hi = orc.search_range(filt, -win_us, win_us)
    out["window_lo"], out["window_hi"] = lo, hi
    wins = [(0, n_samples - 1), (n_samples // 16, pre), (pre - 3, pre + 200), (7, 8)]
    out["td_windows"] = np.array(wins)
    for i, (a, b) in enumerate(wins):
        out[f"td{i}_baseline"] = orc.baseline(x, a, b)
        out[f"td{i}_integral"] = orc.integral(x, FS, a, b)
        out[f"td{i}_maximum"] = orc.maximum(x, a, b)
        out[f"td{i}_minimum"] = orc.minimum(x, a, b)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "written:", n_traces, "x", n_samples)


def make_trigger(fname="golden_trigger_n4096.npz"):
    """Continuous-data trigger (oracle/oftrigger.py): a 120k-sample stream with noise and
    pulses, the filtered / delta-chi2 traces at a few probe points and the triggers for
    two pile-up windows."""
    from oracle import oftrigger as ot
    n, pre, L = 4096, 1500, 120000
    rng = np.random.default_rng(2024)
    tmpl = synth.make_template(n, pre, FS)
    psd = synth.make_psd(n, FS)
    t = ot.OFTrigger(FS, tmpl, psd, pre)
    x = synth.coloured_noise(rng, L // n + 2, psd, FS).reshape(-1)[:L]
    onsets = np.sort(rng.integers(2 * n, L - 3 * n, 9))
    amps = t.resolution * rng.uniform(10, 60, 9)
    for p, a in zip(onsets, amps):
        x[p:p + n] += a * tmpl
    x32 = x.astype(np.float32)
    filt, dchi = t.update_trace(x32.astype(np.float64))
    out = {"template": tmpl, "psd": psd, "fs": FS, "pre": pre, "stream": x32,
           "onsets": onsets, "amps": amps, "probe": np.arange(0, L, 997),
           "filtered_probe": filt[::997], "dchi2_probe": dchi[::997],
           "filtered_max": np.max(np.abs(filt)), "dchi2_max": np.max(dchi)}
    for w in (200, 8192):
        r = t.find_triggers(5.0, pileup_window_samples=w)
        for k in ("trigger_index", "trigger_delta_chi2", "trigger_amplitude"):
            out[f"w{w}_{k}"] = r[k]
        out[f"w{w}_chi2_threshold"] = r["chi2_threshold"]
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "written")


def make_trigger_residual(fname="golden_trigger_residual_n4096.npz"):
    """Dynamic pile-up window and residual pass (oracle/oftrigger.py; oftrigger.py:78-143,
    752-845): a stream with small pulses on the tails of large ones; triggers of the dynamic
    window w(d) = w0 + w1 min(1, d / dref), first / second pass and combined indices of the
    residual pass with a static window, the residual delta-chi2 trace at probe points."""
    from oracle import oftrigger as ot
    n, pre, L = 4096, 2048, 300000
    rng = np.random.default_rng(77)
    tmpl = synth.make_template(n, pre, FS)
    psd = synth.make_psd(n, FS)
    t = ot.OFTrigger(FS, tmpl, psd, pre)
    x = synth.coloured_noise(rng, L // n + 2, psd, FS).reshape(-1)[:L]
    onsets = np.sort(rng.integers(2 * n, L - 3 * n, 10))
    for p in onsets:
        x[p:p + n] += t.resolution * rng.uniform(20, 80) * tmpl
    extra = []
    for p in onsets[::2]:
        q = int(p + rng.integers(n // 8, n // 3))
        x[q:q + n] += t.resolution * rng.uniform(12, 25) * tmpl
        extra.append(q)
    x32 = x.astype(np.float32)
    x64 = x32.astype(np.float64)
    t.update_trace(x64)
    dref = float(np.max(t.delta_chi2))
    w0, w1 = 50.0, 0.4 * n
    fn = lambda d: w0 + w1 * min(1.0, d / dref)
    out = {"template": tmpl, "psd": psd, "fs": FS, "pre": pre, "stream": x32, "onsets": onsets,
           "extra": np.asarray(extra), "dyn_w0": w0, "dyn_w1": w1, "dyn_dref": dref,
           "dchi2_max": dref, "filtered_max": np.max(np.abs(t.filtered))}
    r = t.find_triggers(6.0, dynamic_function=fn)
    out["chi2_threshold"] = r["chi2_threshold"]
    for k in ("trigger_index", "trigger_delta_chi2", "trigger_amplitude"):
        out[f"dyn_{k}"] = r[k]
    first, second, residual, combined = t.find_triggers_residual(6.0, x64, pileup_window_samples=n // 2)
    for nm, rr in (("first", first), ("second", second)):
        for k in ("trigger_index", "trigger_delta_chi2", "trigger_amplitude"):
            out[f"res_{nm}_{k}"] = rr[k]
    out["res_combined_index"] = combined
    out["res_window"] = n // 2
    out["residual_probe"] = residual[::499]
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "written")


if __name__ == "__main__":
    make_trigger()
    make_trigger_residual()
    make(4096, 32, seed=41, fname="golden_n4096.npz")
    make(4096, 8, seed=42, pre=1000, fname="golden_n4096_pre1000.npz")
    make(32768, 6, seed=43, fname="golden_n32768.npz")
    make(25000, 4, seed=44, pre=12500, fname="golden_n25000.npz")
