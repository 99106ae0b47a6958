"""Continuous-data optimal-filter trigger: oracle known-answer tests on
the CPU, GPU parity against the oracle (detprocess/core/oftrigger.py:588-679, 884-1035)."""
import numpy as np
import pytest

from detprocess_amd import synth
from oracle import oftrigger as ot

FS = 1.25e6


def _stream(n_samples, pre, L, seed, n_pulses=12, noise=True):
    rng = np.random.default_rng(seed)
    tmpl = synth.make_template(n_samples, pre, FS)
    psd = synth.make_psd(n_samples, FS)
    trig = ot.OFTrigger(FS, tmpl, psd, pre)
    x = np.zeros(L)
    if noise:
        # coloured noise consistent with the PSD, generated block-wise and stitched
        nb = L // n_samples + 2
        x += synth.coloured_noise(rng, nb, psd, FS).reshape(-1)[:L]
    onsets = np.sort(rng.integers(2 * n_samples, L - 3 * n_samples, n_pulses))
    amps = trig.resolution * rng.uniform(8, 80, n_pulses)
    for p, a in zip(onsets, amps):
        x[p:p + n_samples] += a * tmpl
    return tmpl, psd, trig, x, onsets, amps


def test_oracle_known_answers():
    """Noiseless pulses: amplitude exact, index = onset + pretrigger + 1 (see the oracle's
    header on the one-sample offset of the in-tree arithmetic), pile-up merging by window,
    edge padding, sigma -> chi2 threshold."""
    n, pre, L = 4096, 1500, 80000
    tmpl = synth.make_template(n, pre, FS)
    psd = synth.make_psd(n, FS)
    t = ot.OFTrigger(FS, tmpl, psd, pre)
    x = np.zeros(L)
    pos = [10000, 30000, 30900, 50000]
    amps = [1e-7, 2e-7, 0.5e-7, 3e-7]
    for p, a in zip(pos, amps):
        x[p:p + n] += a * tmpl
    filt, dchi = t.update_trace(x)
    assert np.all(dchi[:n] == 0) and np.all(dchi[-(n) + 1:] == 0)
    r = t.find_triggers(5.0, pileup_window_samples=2000)
    assert list(r["trigger_index"]) == [10000 + pre + 1, 30000 + pre + 1, 50000 + pre + 1]
    assert r["trigger_amplitude"][0] == pytest.approx(1e-7, rel=1e-9)
    assert r["trigger_amplitude"][2] == pytest.approx(3e-7, rel=1e-9)
    assert np.allclose(r["trigger_delta_chi2"], r["trigger_amplitude"] ** 2 * t.w)
    # a window shorter than the gap between the two piled-up excursions separates them
    r2 = t.find_triggers(5.0, pileup_window_samples=0)
    assert len(r2["trigger_index"]) >= 4
    assert ot.OFTrigger.chi2_threshold(5.0) == pytest.approx(25.0, rel=1e-6)
    assert ot.OFTrigger.chi2_threshold(30.0) == 900.0
    # edge exclusion keeps triggers strictly inside (oftrigger.py:851-880)
    r3 = t.find_triggers(5.0, pileup_window_samples=2000, edge_exclusion_msec=10.0)
    assert list(r3["trigger_index"]) == [30000 + pre + 1, 50000 + pre + 1]


def test_oracle_time_domain_equals_frequency_domain_of():
    """At a pulse the trigger's filtered trace is the of1x1 amplitude at that lag: the FIR
    filter and the per-event frequency-domain filter are the same estimator."""
    from oracle import of1x1 as orc
    n, pre = 4096, 2048
    tmpl, psd, t, x, onsets, amps = _stream(n, pre, 12 * 4096, seed=3, n_pulses=1, noise=False)
    filt, _ = t.update_trace(x)
    p = int(onsets[0])
    f = orc.OFFilter(tmpl, psd, FS, pre)
    r = orc.of1x1_nodelay(f, x[p:p + n])
    assert filt[p + n // 2 + 1] == pytest.approx(r["amp"], rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n,pre,L", [(4096, 1500, 300000), (32768, 16384, 1500000), (1000, 300, 50001)])
def test_gpu_trigger_vs_oracle(n, pre, L):
    import torch
    from detprocess_amd import OptimumFilterTrigger
    tmpl, psd, t, x, onsets, amps = _stream(n, pre, L, seed=11)
    x32 = x.astype(np.float32)
    filt, dchi = t.update_trace(x32.astype(np.float64))
    g = OptimumFilterTrigger("chanA", FS, tmpl, psd, pre)
    assert g.get_resolution()[0] == pytest.approx(t.resolution, rel=1e-12)
    assert g.get_norm() == pytest.approx(t.norm_td, rel=1e-9)
    g.update_trace(x32)
    gf = g.get_filtered_trace()[0].astype(np.float64)
    gd = g.get_filtered_delta_chi2().astype(np.float64)
    scale = np.max(np.abs(filt))
    assert np.max(np.abs(gf - filt)) <= 2e-5 * scale
    assert np.all(gd[:n] == 0) and np.all(gd[L - n + 1:] == 0)
    assert np.max(np.abs(gd - dchi)) <= 4e-5 * np.max(dchi)
    for window in (0, 200, 2 * n):
        ref = t.find_triggers(5.0, pileup_window_samples=window)
        g.find_triggers(5.0, pileup_window_samples=window)
        td = g.get_trigger_data()["chanA"]
        thr = ref["chi2_threshold"]
        assert g.get_chi2_threshold() == pytest.approx(thr, rel=1e-12)
        # triggers clear of the threshold must agree exactly; samples within fp32 rounding
        # of the threshold may fall on either side
        margin = 1e-3 * thr + 4e-5 * np.max(dchi)
        ri = {int(i): (d, a) for i, d, a in zip(ref["trigger_index"], ref["trigger_delta_chi2"],
                                                 ref["trigger_amplitude"])}
        gi = {int(i): (d, a) for i, d, a in zip(td["trigger_index"], td["trigger_delta_chi2"],
                                                 td["trigger_amplitude"])}
        clear_ref = {i for i, (d, a) in ri.items() if d > thr + margin}
        clear_gpu = {i for i, (d, a) in gi.items() if d > thr + margin}
        if window >= 200:
            assert clear_ref <= set(gi) and clear_gpu <= set(ri), (window, clear_ref ^ clear_gpu)
        common = set(ri) & set(gi)
        assert len(common) >= 0.9 * max(len(ri), 1)
        for i in common:
            assert gi[i][1] == pytest.approx(ri[i][1], rel=2e-5, abs=2e-5 * scale)
            assert gi[i][0] == pytest.approx(ri[i][0], rel=1e-4, abs=margin)
        assert td["trigger_index_chanA"] == td["trigger_index"]
        assert td["trigger_type"] == [4] * len(td["trigger_index"])
    # every injected pulse is found one sample after onset + pretrigger (static 2N window)
    found = np.array(td["trigger_index"])
    for p, a in zip(onsets, amps):
        near = found[np.abs(found - (p + pre + 1)) <= 3]
        isolated = np.min(np.abs(onsets[onsets != p] - p)) > 3 * n if len(onsets) > 1 else True
        if isolated:
            assert len(near) == 1
    # device-resident input and int16 input give the same filtered trace
    g.update_trace(torch.as_tensor(x32, device="cuda:0"))
    assert np.array_equal(g.get_filtered_trace()[0], gf.astype(np.float32))
    sc = float(np.max(np.abs(x32))) / 30000.0
    adc = np.round(x32 / sc).astype(np.int16)
    g.update_trace(adc, adc_scale=sc)
    t.update_trace(adc.astype(np.float64) * np.float32(sc))
    assert np.max(np.abs(g.get_filtered_trace()[0] - t.filtered)) <= 2e-5 * scale


def _load_trigger_golden():
    from util import load_golden
    return load_golden("golden_trigger_n4096.npz")


def test_oracle_reproduces_trigger_golden():
    g = _load_trigger_golden()
    t = ot.OFTrigger(float(g["fs"]), g["template"], g["psd"], int(g["pre"]))
    filt, dchi = t.update_trace(g["stream"].astype(np.float64))
    assert np.allclose(filt[::997], g["filtered_probe"], rtol=1e-12, atol=0)
    assert np.allclose(dchi[::997], g["dchi2_probe"], rtol=1e-12, atol=0)
    for w in (200, 8192):
        r = t.find_triggers(5.0, pileup_window_samples=w)
        assert np.array_equal(r["trigger_index"], g[f"w{w}_trigger_index"])
        assert np.allclose(r["trigger_amplitude"], g[f"w{w}_trigger_amplitude"], rtol=1e-12)
    # every isolated injected pulse is among the triggers (one sample late, see the oracle)
    found = [int(i) for i in g["w200_trigger_index"]]
    onsets = [int(p) for p in g["onsets"]]
    n = g["template"].shape[0]
    for p in onsets:
        if min(abs(p - o) for o in onsets if o != p) > 2 * n:
            assert any(abs(p + int(g["pre"]) + 1 - f) <= 2 for f in found)


@pytest.mark.gpu
def test_gpu_trigger_reproduces_golden():
    from detprocess_amd import OptimumFilterTrigger
    g = _load_trigger_golden()
    fs, pre = float(g["fs"]), int(g["pre"])
    trig = OptimumFilterTrigger("chanA", fs, g["template"], g["psd"], pre)
    trig.update_trace(g["stream"])
    f = trig.get_filtered_trace()[0].astype(np.float64)
    d = trig.get_filtered_delta_chi2().astype(np.float64)
    assert np.max(np.abs(f[::997] - g["filtered_probe"])) <= 2e-5 * float(g["filtered_max"])
    assert np.max(np.abs(d[::997] - g["dchi2_probe"])) <= 4e-5 * float(g["dchi2_max"])
    for w in (200, 8192):
        trig.find_triggers(5.0, pileup_window_samples=w)
        td = trig.get_trigger_data()["chanA"]
        thr = float(g[f"w{w}_chi2_threshold"])
        margin = 1e-3 * thr + 4e-5 * float(g["dchi2_max"])
        want = {int(i): (dc, a) for i, dc, a in zip(g[f"w{w}_trigger_index"],
                                                    g[f"w{w}_trigger_delta_chi2"],
                                                    g[f"w{w}_trigger_amplitude"])}
        got = {int(i): (dc, a) for i, dc, a in zip(td["trigger_index"], td["trigger_delta_chi2"],
                                                   td["trigger_amplitude"])}
        clear = {i for i, (dc, a) in want.items() if dc > thr + margin}
        assert clear <= set(got)
        assert {i for i, (dc, a) in got.items() if dc > thr + margin} <= set(want)
        for i in set(want) & set(got):
            assert got[i][1] == pytest.approx(want[i][1], rel=2e-5, abs=2e-5 * float(g["filtered_max"]))


# ------------------------------------------------------------------ N channels x M amplitudes
def _stream_nxm(n, pre, C, M, L, seed, n_pulses=10):
    from test_ofnxm import make_csd, make_templates
    rng = np.random.default_rng(seed)
    t = make_templates(n, pre, C, M)
    csd = make_csd(n, C)
    trig = ot.OFTriggerNxM(FS, t, csd, pre)
    x = np.zeros((C, L))
    nb = L // n + 2
    for a in range(C):
        x[a] = synth.coloured_noise(rng, nb, csd[a, a].real, FS).reshape(-1)[:L]
    onsets = np.sort(rng.integers(2 * n, L - 3 * n, n_pulses))
    amps = trig.resolution[None, :] * rng.uniform(10, 80, (n_pulses, M))
    for p, a in zip(onsets, amps):
        x[:, p:p + n] += np.einsum("m,amn->an", a, t)
    return t, csd, trig, x, onsets, amps


def test_oracle_nxm_trigger_known_answers():
    """Noise-free N x M pulses: each amplitude vector is recovered at onset + pretrigger + 1
    (the same one-sample offset as the 1 x 1 arithmetic), and N = M = 1 is the 1 x 1 oracle."""
    from test_ofnxm import make_csd, make_templates
    n, pre, C, M, L = 2048, 700, 2, 2, 40000
    t = make_templates(n, pre, C, M)
    csd = make_csd(n, C)
    tr = ot.OFTriggerNxM(FS, t, csd, pre)
    a = np.array([[4e-7, 1e-7], [-2e-7, 3e-7]])
    x = np.zeros((C, L))
    for p, av in zip((8000, 25000), a):
        x[:, p:p + n] += np.einsum("m,amn->an", av, t)
    tr.update_trace(x)
    r = tr.find_triggers(5.0, pileup_window_samples=n)
    assert list(r["trigger_index"]) == [8000 + pre + 1, 25000 + pre + 1]
    assert np.allclose(r["trigger_amplitudes"], a, rtol=2e-3, atol=1e-3 * np.abs(a).max())
    # the delta chi2 at the pulse is the chi2 reduction A^T P A
    want = np.einsum("pi,ij,pj->p", a, tr.w_matrix, a)
    assert np.allclose(r["trigger_delta_chi2"], want, rtol=5e-3)
    tmpl = synth.make_template(n, pre, FS)
    psd = synth.make_psd(n, FS)
    t1 = ot.OFTrigger(FS, tmpl, psd, pre)
    tn = ot.OFTriggerNxM(FS, tmpl[None, None], psd[None, None], pre)
    rng = np.random.default_rng(0)
    y = synth.coloured_noise(rng, 12, psd, FS).reshape(-1)
    y[9000:9000 + n] += 30 * t1.resolution * tmpl
    f1, d1 = t1.update_trace(y)
    fn, dn = tn.update_trace(y[None, :])
    assert np.allclose(fn[0], f1, rtol=1e-9, atol=1e-12 * np.abs(f1).max())
    assert np.allclose(dn, d1, rtol=1e-9, atol=1e-12 * d1.max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,pre,C,M,L", [(4096, 1500, 2, 2, 300000), (2048, 1024, 3, 2, 120001),
                                         (8192, 4096, 2, 1, 400000)])
def test_gpu_nxm_trigger_vs_oracle(n, pre, C, M, L):
    import torch
    from detprocess_amd import OptimumFilterTrigger
    t, csd, tr, x, onsets, amps = _stream_nxm(n, pre, C, M, L, seed=n + C)
    x32 = x.astype(np.float32)
    filt, dchi = tr.update_trace(x32.astype(np.float64))
    name = "|".join("abc"[:C])
    g = OptimumFilterTrigger(list("abc"[:C]), FS, t, csd, pre)
    assert np.allclose(g.get_resolution(), tr.resolution, rtol=1e-10)
    assert g.get_phi().shape == (C, M, n)
    assert np.allclose(g.get_phi(), tr.phi_td, rtol=1e-9, atol=1e-12 * np.abs(tr.phi_td).max())
    g.update_trace(x32)
    gf = g.get_filtered_trace().astype(np.float64)
    gd = g.get_filtered_delta_chi2().astype(np.float64)
    assert gf.shape == (M, L)
    scale = np.max(np.abs(filt), axis=1, keepdims=True)
    assert np.all(np.abs(gf - filt) <= 3e-5 * scale)
    assert np.all(gd[:n] == 0) and np.all(gd[L - n + 1:] == 0)
    assert np.max(np.abs(gd - dchi)) <= 6e-5 * np.max(dchi)
    ref = tr.find_triggers(5.0, pileup_window_samples=2 * n)
    g.find_triggers(5.0, pileup_window_samples=2 * n)
    td = g.get_trigger_data()[name]
    thr = ref["chi2_threshold"]
    assert g.get_chi2_threshold() == pytest.approx(thr, rel=1e-12)
    margin = 1e-3 * thr + 6e-5 * np.max(dchi)
    ri = {int(i): (d, a) for i, d, a in zip(ref["trigger_index"], ref["trigger_delta_chi2"],
                                             ref["trigger_amplitudes"])}
    gamp = np.array([td[f"trigger_amplitude_{m}"] for m in range(M)]).T.reshape(-1, M)
    gi = {int(i): (d, a) for i, d, a in zip(td["trigger_index"], td["trigger_delta_chi2"], gamp)}
    assert {i for i, (d, a) in ri.items() if d > thr + margin} <= set(gi)
    assert {i for i, (d, a) in gi.items() if d > thr + margin} <= set(ri)
    common = set(ri) & set(gi)
    assert len(common) >= 0.9 * max(len(ri), 1) and len(common) >= 3
    for i in common:
        assert np.all(np.abs(gi[i][1] - ri[i][1]) <= 3e-5 * scale[:, 0] + 2e-5 * np.abs(ri[i][1]))
        assert gi[i][0] == pytest.approx(ri[i][0], rel=1e-4, abs=margin)
    assert ("trigger_amplitude" in td) == (M == 1)
    assert td[f"trigger_index_{name}"] == td["trigger_index"]
    # isolated injected pulses are found one sample after onset + pretrigger (a pulse stays above
    # threshold for about a trace length either side, and the window merges gaps up to 2 n)
    found = np.array(td["trigger_index"])
    for p in onsets:
        if np.min(np.abs(onsets[onsets != p] - p)) > 6 * n:
            assert np.sum(np.abs(found - (p + pre + 1)) <= 3) == 1
    # device-resident and int16 inputs
    g.update_trace(torch.as_tensor(x32, device="cuda:0"))
    assert np.array_equal(g.get_filtered_trace(), gf.astype(np.float32))
    sc = np.max(np.abs(x32), axis=1) / 30000.0
    adc = np.round(x32 / sc[:, None]).astype(np.int16)
    g.update_trace(adc, adc_scale=sc, adc_offset=np.zeros(C))
    tr.update_trace(adc.astype(np.float64) * sc.astype(np.float32)[:, None])
    assert np.all(np.abs(g.get_filtered_trace() - tr.filtered) <= 3e-5 * scale)
    with pytest.raises(ValueError):
        g.update_trace(x32[:1] if C > 1 else np.zeros((2, L), dtype=np.float32))


@pytest.mark.gpu
def test_trigger_then_features_on_one_int16_stream():
    """The stage before the path feeding the path (triggers.py -> features.py in the reference):
    triggers found on a continuous int16 stream are handed, as trigger indices, to the batched
    feature driver, which cuts the events out of the same stream on the GPU.  Every injected,
    isolated pulse comes back with its amplitude and a fitted delay of one sample (the trigger
    index sits one sample after onset + pretrigger, see oracle/oftrigger.py)."""
    from detprocess_amd import FeatureProcessing, FilterData, OptimumFilterTrigger
    n, pre, L = 4096, 2048, 800000
    tmpl, psd, t, x, onsets, amps = _stream(n, pre, L, seed=6, n_pulses=10)
    scale = float(np.max(np.abs(x))) / 30000.0
    adc = np.round(x / scale).astype(np.int16)[None, :]
    trig = OptimumFilterTrigger("chanA", FS, tmpl, psd, pre)
    trig.update_trace(adc[0], adc_scale=scale)
    trig.find_triggers(8.0, pileup_window_samples=2 * n)
    found = np.array(trig.get_trigger_data()["chanA"]["trigger_index"], dtype=np.int64)
    fd = FilterData()
    fd.set_template("chanA", tmpl, sample_rate=FS, pretrigger_length_samples=pre, tag="default")
    fd.set_psd("chanA", psd, np.fft.fftfreq(n, d=1 / FS), sample_rate=FS, tag="default")
    yaml_text = """
chanA:
    of1x1_constrained:
        run: True
        template_tag: default
        window_min_from_trig_usec: -50
        window_max_from_trig_usec: 50
    baseline:
        run: True
        window_min_from_start_usec: 0
        window_max_from_trig_usec: -500
"""
    fp = FeatureProcessing(yaml_text, fd, ["chanA"], FS, nb_samples=n, nb_pretrigger_samples=pre)
    df = fp.process_adc(adc, found, scale, 0.0)
    assert len(df) == len(found)
    checked = 0
    for p, a in zip(onsets, amps):
        if np.min(np.abs(onsets[onsets != p] - p)) <= 6 * n:
            continue
        hit = np.nonzero(np.abs(found - (p + pre + 1)) <= 2)[0]
        assert len(hit) == 1
        row = df.iloc[hit[0]]
        assert row["amp_of1x1_constrained_chanA"] == pytest.approx(a, rel=0.2, abs=3 * t.resolution)
        assert abs(round(row["t0_of1x1_constrained_chanA"] * FS) + (found[hit[0]] - (p + pre))) <= 1
        checked += 1
    assert checked >= 5


# ------------------------------------------------- dynamic window, residual pass, saturation
def test_dynamic_ranges_of_the_product_equal_the_literal_restatement():
    """detprocess_amd.oftrigger._dynamic_ranges (running maximum, cached window) against the
    oracle's literal restatement of _getchangeslessthandynamicthresh (oftrigger.py:78-143)."""
    from detprocess_amd.oftrigger import _dynamic_ranges
    rng = np.random.default_rng(4)
    fns = [lambda a: 3.0, lambda a: 0.5, lambda a: 2.0 + 0.02 * a, lambda a: 40.0 / (1.0 + a)]
    for trial in range(40):
        m = int(rng.integers(1, 300))
        x = np.sort(rng.choice(5000, size=m, replace=False))
        a = rng.exponential(100.0, size=m)
        for fn in fns:
            assert _dynamic_ranges(x, a, fn) == ot.dynamic_ranges(x, a, fn)
    # nothing above threshold: the reference's helper yields the empty range (0, 0), which
    # find_triggers_once skips (oftrigger.py:997); the product returns no range at all
    assert _dynamic_ranges(np.array([], dtype=np.int64), np.array([]), fns[0]) == []
    assert ot.dynamic_ranges(np.array([], dtype=np.int64), np.array([]), fns[0]) == [(0, 0)]
    # the pile-up window grows with the range maximum: a large excursion swallows what follows
    x = np.array([10, 11, 12, 40, 41, 100])
    a = np.array([30.0, 900.0, 30.0, 35.0, 30.0, 31.0])
    assert _dynamic_ranges(x, a, lambda v: 5.0 if v < 100 else 50.0) == [(0, 5), (5, 6)]
    assert _dynamic_ranges(x, a, lambda v: 5.0) == [(0, 3), (3, 5), (5, 6)]


def test_oracle_residual_pass_finds_the_piled_up_pulse():
    """Known answer for oftrigger.py:752-845: a small pulse on the tail of a large one is
    hidden by a long static pile-up window; after the large pulse's delta-chi2 shape is
    subtracted the re-trigger finds it; a saturation veto on the large pulse keeps it hidden."""
    # (pretrigger = n/2: the reference reads the amplitude and places the subtraction at the
    # stored trigger index, which carries the shift pretrigger - n//2, oftrigger.py:766-815)
    n, pre, L = 4096, 2048, 120000
    tmpl = synth.make_template(n, pre, FS)
    psd = synth.make_psd(n, FS)
    t = ot.OFTrigger(FS, tmpl, psd, pre)
    x = np.zeros(L)
    big, small = 30000, 30700
    x[big:big + n] += 2e-7 * tmpl
    x[small:small + n] += 0.5e-7 * tmpl
    x[80000:80000 + n] += 1e-7 * tmpl
    t.update_trace(x)
    first, second, residual, combined = t.find_triggers_residual(5.0, x, pileup_window_samples=3000)
    assert list(first["trigger_index"]) == [big + pre + 1, 80000 + pre + 1]
    assert abs(int(second["trigger_index"][0]) - (small + pre + 1)) <= 2
    assert len(combined) == 3 and set(first["trigger_index"]) < set(combined)
    assert np.max(residual) < 0.1 * np.max(t.delta_chi2)
    # the first-pass trace is restored
    assert np.max(t.delta_chi2) == pytest.approx((2e-7) ** 2 * t.w, rel=0.05)
    sat = [1.5e-7]
    f2, s2, res2, comb2 = t.find_triggers_residual(5.0, x, pileup_window_samples=3000, saturation=sat)
    assert list(comb2[:2]) == list(first["trigger_index"])
    assert len(comb2) == 2                      # vetoed: nothing subtracted, nothing new found
    assert np.max(res2[big + pre - 50: big + pre + 50]) > 0.9 * (2e-7) ** 2 * t.w


def _pileup_stream(n, pre, L, seed):
    tmpl, psd, t, x, onsets, amps = _stream(n, pre, L, seed=seed, n_pulses=10)
    rng = np.random.default_rng(seed + 1)
    extra = []
    for p in onsets[::2]:                        # a small pulse on the tail of every other one
        q = int(p + rng.integers(n // 8, n // 3))
        if q + n < L:
            x[q:q + n] += t.resolution * rng.uniform(12, 25) * tmpl
            extra.append(q)
    return tmpl, psd, t, x, onsets, np.asarray(extra)


@pytest.mark.gpu
@pytest.mark.parametrize("n,pre,L", [(4096, 1500, 400000), (4096, 2048, 400000),
                                     (32768, 16384, 1800000)])
def test_gpu_dynamic_window_and_residual_vs_oracle(n, pre, L):
    from detprocess_amd import OptimumFilterTrigger
    tmpl, psd, t, x, onsets, extra = _pileup_stream(n, pre, L, seed=21)
    x32 = x.astype(np.float32)
    x64 = x32.astype(np.float64)
    t.update_trace(x64)
    dmax = float(np.max(t.delta_chi2))
    g = OptimumFilterTrigger("chanA", FS, tmpl, psd, pre)
    g.update_trace(x32)
    fn = lambda d: 50.0 + 0.4 * n * min(1.0, d / dmax)           # window grows with the maximum
    # ---- dynamic pile-up window (oftrigger.py:78-143, 982-986)
    ref = t.find_triggers(6.0, dynamic_function=fn)
    with pytest.raises(ValueError, match="dynamic_threshold_function"):
        g.find_triggers_once(6.0, dynamic=True)
    g.find_triggers(6.0, dynamic=True, dynamic_threshold_function=fn)
    td = g.get_trigger_data()["chanA"]
    thr = ref["chi2_threshold"]
    margin = 1e-3 * thr + 4e-5 * dmax
    clear = [int(i) for i, d in zip(ref["trigger_index"], ref["trigger_delta_chi2"]) if d > 2 * thr + margin]
    assert set(clear) <= set(td["trigger_index"])
    assert abs(len(td["trigger_index"]) - len(ref["trigger_index"])) <= max(2, len(ref["trigger_index"]) // 10)
    ri = dict(zip(ref["trigger_index"].tolist(), ref["trigger_amplitude"]))
    for i, a, d in zip(td["trigger_index"], td["trigger_amplitude"], td["trigger_delta_chi2"]):
        if i in ri:
            assert a == pytest.approx(ri[i], rel=2e-5, abs=2e-5 * np.max(np.abs(t.filtered)))
    assert td["trigger_pileup_window"] == [0] * len(td["trigger_index"])
    # ---- residual pass (oftrigger.py:752-845), static window long enough to hide the extras
    win = n // 2
    first, second, residual, combined = t.find_triggers_residual(6.0, x64, pileup_window_samples=win)
    out = g.find_triggers(6.0, pileup_window_samples=win, residual=True, return_trigger_data=True)
    o_first, o_dchi, o_second, o_res = out
    gi1 = o_first["chanA"]["trigger_index"]
    gi2 = o_second["chanA"]["trigger_index"]
    c1 = {int(i) for i, d in zip(first["trigger_index"], first["trigger_delta_chi2"]) if d > 2 * thr}
    c2 = {int(i) for i, d in zip(second["trigger_index"], second["trigger_delta_chi2"]) if d > 2 * thr}
    assert c1 <= set(gi1) and c2 <= set(gi2)
    if pre == n // 2:
        assert len(set(gi2) - set(gi1)) >= 1                # the hidden pulses come out
    assert np.max(np.abs(o_res.astype(np.float64) - residual)) <= 1e-4 * dmax
    assert np.max(np.abs(o_dchi.astype(np.float64) - t.delta_chi2)) <= 4e-5 * dmax
    comb = g.get_trigger_data()["chanA"]
    assert comb["trigger_index"][:len(gi1)] == gi1          # first pass first, new ones appended
    assert set(comb["trigger_index"]) == set(gi1) | set(gi2)
    assert len(comb["trigger_index"]) == len(set(comb["trigger_index"]))
    for key in ("trigger_delta_chi2", "trigger_time", "trigger_amplitude", "trigger_type",
                "trigger_channel", "trigger_index_chanA"):
        assert len(comb[key]) == len(comb["trigger_index"]), key
    # the delta-chi2 trace on the device is the first-pass one again
    assert np.array_equal(g.get_filtered_delta_chi2(), o_dchi)
    # ---- saturation veto: with every large pulse "saturated" nothing is subtracted
    lp = ot.lowpass_50khz(x64, FS)
    level = 0.3 * float(np.max(lp))
    f3, s3, r3, comb3 = t.find_triggers_residual(6.0, x64, pileup_window_samples=win, saturation=[level])
    out3 = g.find_triggers(6.0, pileup_window_samples=win, residual=True,
                           saturation_amplitudes_LPF_50kHz=[level], return_trigger_data=True)
    assert np.max(np.abs(out3[3].astype(np.float64) - r3)) <= 1e-4 * dmax
    big = [int(i) for i in f3["trigger_index"] if np.max(lp[i - n // 4: i + n // 4]) > level]
    assert big, "the test stream must contain saturated pulses"
    assert np.max(r3[big]) > 0.5 * np.min(t.delta_chi2[big])       # the vetoed pulses stay


@pytest.mark.gpu
def test_gpu_residual_pass_two_channels_two_amplitudes():
    """N x M (2 x 2): the pulse table G_ab[z] against the oracle's per-trigger construction."""
    from detprocess_amd import OptimumFilterTrigger
    from test_ofnxm import make_csd, make_templates
    n, pre, L = 4096, 2048, 200000
    tm = make_templates(n, pre, 2, 2)
    csd = make_csd(n, 2)
    t = ot.OFTriggerNxM(FS, tm, csd, pre)
    rng = np.random.default_rng(6)
    x = 2e-12 * rng.standard_normal((2, L))
    for p, a, b in ((20000, 40.0, 10.0), (20500, 4.0, 3.0), (90000, 25.0, -8.0), (150000, 6.0, 30.0)):
        for c in range(2):
            x[c, p:p + n] += t.resolution[0] * a * tm[c, 0] + t.resolution[1] * b * tm[c, 1]
    x32 = x.astype(np.float32)
    x64 = x32.astype(np.float64)
    t.update_trace(x64)
    dmax = float(np.max(t.delta_chi2))
    g = OptimumFilterTrigger(["A", "B"], FS, tm, csd, pre, trigger_name="AB")
    g.update_trace(x32)
    first, second, residual, combined = t.find_triggers_residual(6.0, x64, pileup_window_samples=2000)
    out = g.find_triggers(6.0, pileup_window_samples=2000, residual=True, return_trigger_data=True)
    thr = first["chi2_threshold"]
    c1 = {int(i) for i, d in zip(first["trigger_index"], first["trigger_delta_chi2"]) if d > 2 * thr}
    assert c1 <= set(out[0]["AB"]["trigger_index"])
    assert np.max(np.abs(out[3].astype(np.float64) - residual)) <= 2e-4 * dmax
    c2 = {int(i) for i, d in zip(second["trigger_index"], second["trigger_delta_chi2"]) if d > 2 * thr}
    assert c2 <= set(out[2]["AB"]["trigger_index"])


@pytest.mark.parametrize("shape", ["1x1", "2x2"])
def test_pulse_table_reproduces_the_per_trigger_construction(shape):
    """detprocess_amd.oftrigger.pulse_table (host precompute of the residual pass): for random
    amplitudes sum_ab A_a A_b G_ab[z] equals the delta-chi2 pulse the reference builds per trigger
    (oftrigger.py:793-809), as restated by the oracle's find_triggers_residual."""
    from detprocess_amd.oftrigger import pulse_table
    from scipy.signal import oaconvolve
    n, pre = 1024, 512
    rng = np.random.default_rng(2)
    if shape == "1x1":
        tmpl = synth.make_template(n, pre, FS)
        t = ot.OFTrigger(FS, tmpl, synth.make_psd(n, FS), pre)
        G = pulse_table(tmpl.reshape(1, 1, n), t.phi_td.reshape(1, 1, n),
                        np.array([[1.0 / t.vscale]]), np.array([[t.w]]))
        for amp in (3e-8, -1.2e-7):
            v = oaconvolve(tmpl * amp, t.phi_td, mode="same") / t.vscale
            assert np.allclose(amp * amp * G[0, 0], v * t.w * v, rtol=1e-10, atol=1e-12 * np.max(v * t.w * v))
        return
    from test_ofnxm import make_csd, make_templates
    tm = make_templates(n, pre, 2, 2)
    t = ot.OFTriggerNxM(FS, tm, make_csd(n, 2), pre)
    G = pulse_table(tm, t.phi_td, t.iw_matrix / FS, t.w_matrix)
    for _ in range(3):
        amps = rng.normal(size=2) * t.resolution * 30
        trig_trace = sum(tm[:, m, :] * amps[m] for m in range(2))
        v_td = np.stack([np.sum(oaconvolve(trig_trace, t.phi_td[theta, :], mode="same", axes=-1), axis=0)
                         for theta in range(2)])
        filt = np.einsum("ij,jz->iz", t.iw_matrix / FS, v_td)
        want = np.einsum("iz,ij,jz->z", filt, t.w_matrix, filt)
        got = np.einsum("a,b,abz->z", amps, amps, G)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-12 * np.max(want))


def _load_residual_golden():
    from util import load_golden
    return load_golden("golden_trigger_residual_n4096.npz")


def test_oracle_reproduces_residual_golden():
    g = _load_residual_golden()
    t = ot.OFTrigger(float(g["fs"]), g["template"], g["psd"], int(g["pre"]))
    x64 = g["stream"].astype(np.float64)
    t.update_trace(x64)
    fn = lambda d: float(g["dyn_w0"]) + float(g["dyn_w1"]) * min(1.0, d / float(g["dyn_dref"]))
    r = t.find_triggers(6.0, dynamic_function=fn)
    assert np.array_equal(r["trigger_index"], g["dyn_trigger_index"])
    first, second, residual, combined = t.find_triggers_residual(
        6.0, x64, pileup_window_samples=int(g["res_window"]))
    assert np.array_equal(first["trigger_index"], g["res_first_trigger_index"])
    assert np.array_equal(second["trigger_index"], g["res_second_trigger_index"])
    assert np.array_equal(combined, g["res_combined_index"])
    assert np.allclose(residual[::499], g["residual_probe"], rtol=1e-10, atol=1e-12 * float(g["dchi2_max"]))
    # the fixture does what it is for: the second pass brings out pulses the first one merged
    new = set(g["res_second_trigger_index"].tolist()) - set(g["res_first_trigger_index"].tolist())
    pre = int(g["pre"])
    assert sum(any(abs(q + pre + 1 - i) <= 3 for i in new) for q in g["extra"]) >= 3


@pytest.mark.gpu
def test_gpu_trigger_reproduces_residual_golden():
    from detprocess_amd import OptimumFilterTrigger
    g = _load_residual_golden()
    fs, pre = float(g["fs"]), int(g["pre"])
    trig = OptimumFilterTrigger("chanA", fs, g["template"], g["psd"], pre)
    trig.update_trace(g["stream"])
    thr, dmax = float(g["chi2_threshold"]), float(g["dchi2_max"])
    fn = lambda d: float(g["dyn_w0"]) + float(g["dyn_w1"]) * min(1.0, d / float(g["dyn_dref"]))
    trig.find_triggers(6.0, dynamic=True, dynamic_threshold_function=fn)
    got = trig.get_trigger_data()["chanA"]["trigger_index"]
    clear = [int(i) for i, d in zip(g["dyn_trigger_index"], g["dyn_trigger_delta_chi2"]) if d > 2 * thr]
    assert set(clear) <= set(got) and abs(len(got) - len(g["dyn_trigger_index"])) <= 2
    out = trig.find_triggers(6.0, pileup_window_samples=int(g["res_window"]), residual=True,
                             return_trigger_data=True)
    for key, o in (("first", out[0]), ("second", out[2])):
        want = {int(i) for i, d in zip(g[f"res_{key}_trigger_index"], g[f"res_{key}_trigger_delta_chi2"])
                if d > 2 * thr}
        assert want <= set(o["chanA"]["trigger_index"]), key
    assert np.max(np.abs(out[3][::499].astype(np.float64) - g["residual_probe"])) <= 1e-4 * dmax
    comb = trig.get_trigger_data()["chanA"]["trigger_index"]
    assert {int(i) for i in g["res_combined_index"]
            if i in set(g["res_first_trigger_index"].tolist())} <= set(comb)
