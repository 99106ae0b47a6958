"""BASELINE configs[3] as stated: 4 channels x 32768 samples, 3 template tags (pulse / glitch /
muon), the full per-channel feature set of the reference's example
(examples/processing/process_example.yaml:109-222, lowchi2_fcutoff 50 kHz), driven through
``FeatureProcessing`` -- the same YAML text ``bench.py --config 3`` runs -- and checked column by
column against the oracle.  Every plan must stay on the FUSED kernel.  The same at the example's
own trace length, 25000 samples (process_example.yaml:93; k_fused25)."""

import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from detprocess_amd import synth   # noqa: E402

pytestmark = pytest.mark.gpu
FS = 1.25e6


@pytest.mark.parametrize("n", [32768, 25000])
def test_four_channels_three_tags_full_yaml_set_on_the_fused_kernel(n, monkeypatch):
    import bench
    from detprocess_amd import FeatureProcessing
    from oracle import of1x1 as orc
    from util import AMP_ATOL_SIGMA, AMP_RTOL, CHI_ATOL_CHI0, CHI_RTOL
    monkeypatch.setattr(bench, "N_SAMPLES", n)
    pre, B = n // 2, 10
    fd = bench.filter_data3(pre)
    J = synth.make_psd(n, FS)
    filts = {t: orc.OFFilter(synth.make_template(n, pre, FS, t), J, FS, pre) for t in bench.TAGS3}
    rng = np.random.default_rng(3)
    ev = np.empty((B, 4, n), dtype=np.float32)
    for c in range(4):                 # a different mix of pulse shapes on every channel
        kinds = rng.integers(0, 3, B)
        for b in range(B):
            tag = bench.TAGS3[kinds[b]]
            x, _, _ = synth.make_traces(1, synth.make_template(n, pre, FS, tag), J, FS,
                                        filts[tag].ampres, seed=100 * c + b, max_delay=300)
            ev[b, c] = x[0]
    fp = FeatureProcessing(bench.yaml_config3(), fd, bench.CHANNELS3, FS)
    valid = np.ones(B, dtype=np.uint8)
    valid[7] = 0
    df = fp.process(ev, valid=valid)
    plans = fp.plans()
    assert len(plans) == 4 and {p.engine for p in plans.values()} == {"fused"}
    ok = valid.astype(bool)
    assert (df.iloc[7] == -999999.0).all()
    ncol = 3 * (3 + 4 + 7) + 5 + 5
    assert len(df.columns) == 4 * ncol

    def close(got, want, rtol, atol, what):
        got, want = np.asarray(got)[ok], np.asarray(want)[ok]
        assert np.all(np.abs(got - want) <= rtol * np.abs(want) + atol), what

    for c, ch in enumerate(bench.CHANNELS3):
        x = ev[:, c, :].astype(np.float64)
        for tag, f in filts.items():
            chi0 = None
            runs = {
                "nodelay": orc.process_events(f, x, "nodelay", lowchi2_fcutoff=50000.0),
                "unconstrained": orc.process_events(f, x, "unconstrained"),
                "constrained": orc.process_events(f, x, "constrained", lowchi2_fcutoff=50000.0,
                                                  window_min_from_trig_usec=-100,
                                                  window_max_from_trig_usec=100)}
            chi0 = runs["unconstrained"]["chi2nopulse"]
            for algo, r in runs.items():
                name = f"of1x1_{algo}_{tag}_{ch}"
                close(df[f"amp_{name}"], r["amp"], AMP_RTOL, AMP_ATOL_SIGMA * f.ampres, name)
                close(df[f"chi2_{name}"], r["chi2"], CHI_RTOL, CHI_ATOL_CHI0 * chi0[ok], name)
                close(df[f"lowchi2_{name}"], r["lowchi2"], CHI_RTOL, CHI_ATOL_CHI0 * chi0[ok], name)
                if algo == "nodelay":
                    assert f"t0_{name}" not in df.columns
                    continue
                assert np.array_equal(np.round(np.asarray(df[f"t0_{name}"])[ok] * FS),
                                      (r["index"] - pre)[ok]), name
                if algo == "constrained":
                    close(df[f"chi2nopulse_{name}"], r["chi2nopulse"], CHI_RTOL, 0.0, name)
                    close(df[f"ampres_{name}"], np.full(B, f.ampres), 1e-6, 0.0, name)
                    rel = 2e-5 + (AMP_RTOL * np.abs(r["amp"]) + AMP_ATOL_SIGMA * f.ampres) \
                        / np.maximum(np.abs(r["amp"]), 1e-300)
                    assert np.all(np.abs(np.asarray(df[f"timeres_{name}"]) - r["timeres"])[ok]
                                  <= (rel * r["timeres"])[ok]), name
        scale = np.abs(x).max()
        for col, kw, fn in (
                ("baseline", dict(window_min_from_start_usec=0, window_max_from_trig_usec=-2000),
                 orc.baseline),
                ("baseline_end", dict(window_min_from_trig_usec=2000, window_max_to_end_usec=0),
                 orc.baseline),
                ("maximum", dict(window_min_from_trig_usec=-500, window_max_from_trig_usec=500),
                 orc.maximum),
                ("minimum", dict(window_min_from_trig_usec=-500, window_max_from_trig_usec=500),
                 orc.minimum)):
            lo, hi = orc.get_window_indices(n, pre, FS, **kw)
            want = fn(x, lo, hi)
            if col in ("maximum", "minimum"):
                assert np.array_equal(np.asarray(df[f"{col}_{ch}"])[ok], want[ok]), col
            else:
                close(df[f"{col}_{ch}"], want, 1e-5, 2e-6 * scale, col)
        lo, hi = orc.get_window_indices(n, pre, FS, window_min_from_trig_usec=-10,
                                        window_max_from_trig_usec=500)
        close(df[f"integral_{ch}"], orc.integral(x, FS, lo, hi), 1e-5,
              2e-6 * scale * (hi - lo) / FS, "integral")
        pa = orc.psd_amp(x, FS, [[45.0, 75.0], [300.0, 500.0], [350.0, 450.0], [150, 250], [250, 350]])
        for nm, v in pa.items():
            close(df[f"psd_amp_{nm}_{ch}"], v, 1e-5, 0.0, nm)


def test_device_resident_rows_equal_the_dataframe():
    """``process_device`` (what bench.py --config 3 times) returns the rows ``process`` turns into
    the DataFrame."""
    import torch
    import bench
    from detprocess_amd import FeatureProcessing
    n, pre, B = 32768, 16384, 5
    fd = bench.filter_data3(pre)
    J = synth.make_psd(n, FS)
    x, _, _ = synth.make_traces(4 * B, synth.make_template(n, pre, FS), J, FS, 1e-9, seed=2)
    ev = x.reshape(B, 4, n).astype(np.float32)
    fp = FeatureProcessing(bench.yaml_config3(), fd, bench.CHANNELS3, FS)
    df = fp.process(ev)
    res = fp.process_device(torch.as_tensor(ev, device="cuda:0"))
    cols = fp.device_columns()
    assert set(res) == set(cols)
    seen = 0
    for key, mat in res.items():
        m = mat.cpu().numpy().astype(np.float64)
        for name, off in cols[key]:
            assert np.array_equal(m[:, off], np.asarray(df[name])), name
            seen += 1
    assert seen == len(df.columns)


@pytest.mark.parametrize("n,fcut,want", [(25000, 100000.0, "rocfft"), (25000, 50000.0, "fused"),
                                         (12500, 50000.0, "fused"), (12500, 100000.0, "rocfft"),
                                         (32768, 150000.0, "fused"), (32768, 200000.0, "rocfft")])
def test_plans_beyond_the_stashed_bins_are_compiled_for_rocfft(n, fcut, want, monkeypatch):
    """A lowchi2 cut-off (or psd_amp band) beyond the bins the fused kernel of that trace length keeps
    (32768: 4096, 25000: 1250, 12500: 625; process.FUSED_MAX_BIN) picks the ROCFFT engine when the plan
    is compiled -- with engine='fused' as with 'auto' -- instead of failing on every call; the result
    equals the oracle either way."""
    import bench
    from detprocess_amd import FeatureProcessing
    from detprocess_amd.process import engine_bin_limit
    from oracle import of1x1 as orc
    from util import check_search
    monkeypatch.setattr(bench, "N_SAMPLES", n)
    pre, B = n // 2, 6
    nbins = int(np.floor(fcut * n / FS)) + 1
    assert (nbins > engine_bin_limit(n)) == (want == "rocfft")
    fd = bench.filter_data3(pre)
    yaml = ("chA:\n    of1x1_unconstrained:\n        run: True\n        template_tag: pulse\n"
            f"        csd_tag: default\n        lowchi2_fcutoff: {fcut}\n")
    J = synth.make_psd(n, FS)
    tp = synth.make_template(n, pre, FS, "pulse")
    filt = orc.OFFilter(tp, J, FS, pre)
    x, _, _ = synth.make_traces(B, tp, J, FS, filt.ampres, seed=9, max_delay=300)
    ev = np.zeros((B, 4, n), dtype=np.float32)
    ev[:, 0] = x
    for engine in ("fused", "auto"):
        fp = FeatureProcessing(yaml, fd, bench.CHANNELS3, FS, engine=engine)
        df = fp.process(ev)
        assert {p.engine for p in fp.plans().values()} == {want}, engine
        ref = orc.process_events(filt, ev[:, 0].astype(np.float64), "unconstrained", fcut)
        out = np.zeros((B, 8))
        for j, k in enumerate(("amp", "t0", "chi2", "lowchi2")):
            out[:, j] = df[f"{k}_of1x1_unconstrained_chA"]
        out[:, 4], out[:, 5], out[:, 6], out[:, 7] = ref["chi2nopulse"], ref["ampres"], ref["timeres"], ref["index"]
        check_search(out, 0, ref, "", filt.ampres, FS, f"{n} / {fcut} / {engine}", lowchi2_fcutoff=fcut)
