"""N-channel x M-template optimal filter (algorithms.py:141-274):
oracle known-answer tests and host precompute on the CPU, GPU parity through the C ABI."""
import numpy as np
import pytest

from detprocess_amd import synth
from oracle import of1x1 as o1
from oracle import ofnxm as onm

FS = 1.25e6
# fp64 -> fp32 tolerances, as for of1x1 (tests/util.py): amplitudes 2e-5 relative plus 1e-4 of
# their resolution; chi2 2e-5 relative plus 2e-6 of chi2_0; the time bin exact
AMP_RTOL, AMP_ATOL_SIGMA, CHI_RTOL, CHI_ATOL_CHI0 = 2e-5, 1e-4, 2e-5, 2e-6


def make_csd(n, n_chan, rho=0.3):
    """Hermitian positive-definite two-sided CSD [C, C, n] with complex off-diagonal terms."""
    J = synth.make_psd(n, FS)
    sgn = np.sign(np.fft.fftfreq(n, d=1 / FS))
    scale = np.array([1.0, 1.5, 0.7, 2.0])[:n_chan]
    csd = np.zeros((n_chan, n_chan, n), dtype=np.complex128)
    for a in range(n_chan):
        csd[a, a] = J * scale[a]
        for b in range(a + 1, n_chan):
            csd[a, b] = rho * J * np.sqrt(scale[a] * scale[b]) * np.exp(0.4j * (b - a) * sgn)
            csd[b, a] = np.conj(csd[a, b])
    return csd


def make_templates(n, pre, n_chan, n_tmpl):
    kinds = ["pulse", "muon", "glitch", "pulse"]
    share = np.array([[1.0, 0.2, 0.5, 0.1], [0.4, 1.0, -0.3, 0.6], [0.7, -0.5, 1.0, 0.2],
                      [0.2, 0.3, 0.4, 1.0]])
    t = np.zeros((n_chan, n_tmpl, n))
    for m in range(n_tmpl):
        shape = synth.make_template(n, pre, FS, kinds[m])
        if m == 3:
            shape = np.roll(shape, 40)
        for a in range(n_chan):
            t[a, m] = share[a, m] * shape
    return t


def make_events(B, templates, csd, ampres, seed, max_delay):
    rng = np.random.Generator(np.random.PCG64(seed))
    C, M, n = templates.shape
    ev = np.zeros((B, C, n))
    for a in range(C):
        ev[:, a] = synth.coloured_noise(rng, B, csd[a, a].real, FS)
    amps = np.where(rng.random((B, M)) < 0.6,
                    ampres * np.exp(rng.uniform(np.log(3), np.log(300), (B, M))), 0.0)
    d = rng.integers(-max_delay, max_delay + 1, B)
    for e in range(B):
        for m in range(M):
            ev[e] += amps[e, m] * np.roll(templates[:, m], d[e], axis=-1)
    return ev, amps, d


# ------------------------------------------------------------------ CPU: oracle + host tables
def test_oracle_reduces_to_of1x1():
    n, pre = 1024, 400
    tmpl = synth.make_template(n, pre, FS)
    J = synth.make_psd(n, FS)
    f1 = o1.OFFilter(tmpl, J, FS, pre)
    fn = onm.NxMFilter(tmpl[None, None], J[None, None], FS, pre)
    assert np.isclose(fn.P[0, 0], f1.norm, rtol=1e-12)
    x, _, _ = synth.make_traces(6, tmpl, J, FS, f1.ampres, seed=3, max_delay=100)
    r1 = o1.process_events(f1, x, "constrained", window_min_from_trig_usec=-100,
                           window_max_from_trig_usec=100)
    rn = onm.process_events(fn, x[:, None, :], window_min_from_trig_usec=-100,
                            window_max_from_trig_usec=100)
    assert np.array_equal(rn["index"], r1["index"])
    assert np.allclose(rn["amps"][:, 0], r1["amp"], rtol=1e-10)
    assert np.allclose(rn["chi2"], r1["chi2"], rtol=1e-9)
    r0 = o1.process_events(f1, x, "nodelay")
    assert np.allclose(rn["amps_nodelay"][:, 0], r0["amp"], rtol=1e-10)
    assert np.allclose(rn["chi2_nodelay"], r0["chi2"], rtol=1e-9)


def test_oracle_known_answer_noise_free():
    """A noise-free sum of the templates shifted by d is fitted exactly: amplitudes recovered,
    t0 = d / fs, chi2 = 0 (up to rounding of chi2_0)."""
    n, pre, C, M = 2048, 1000, 3, 2
    t = make_templates(n, pre, C, M)
    filt = onm.NxMFilter(t, make_csd(n, C), FS, pre)
    a = np.array([3e-7, -1.2e-7])
    for d in (0, 17, -40):
        x = np.einsum("m,amn->an", a, np.roll(t, d, axis=-1))
        r = onm.fit(filt, x)
        assert r["index"] == pre + d and np.isclose(r["t0"], d / FS)
        assert np.allclose(r["amps"], a, rtol=1e-9)
        assert abs(r["chi2"]) < 1e-9 * r["chi2_0"]
    # weight matrix: symmetric positive definite, resolutions from its inverse
    assert np.all(np.linalg.eigvalsh(filt.P) > 0)
    assert np.allclose(filt.ampres ** 2, np.diag(np.linalg.inv(filt.P)))
    # the window restricts the search; lgc_outside_window searches the complement
    x = np.einsum("m,amn->an", a, np.roll(t, 300, axis=-1))
    r_in = onm.fit(filt, x, window_min_from_trig_usec=-100, window_max_from_trig_usec=100)
    lo, hi = o1.search_range(filt, -100, 100)
    assert lo <= r_in["index"] < hi
    r_out = onm.fit(filt, x, window_min_from_trig_usec=-100, window_max_from_trig_usec=100,
                    lgc_outside_window=True)
    assert r_out["index"] == pre + 300


def test_host_tables_match_oracle():
    from detprocess_amd.ofnxm import build_nxm_filter, nxm_search_range
    n, pre, C, M = 1000, 300, 2, 3
    t = make_templates(n, pre, C, M)
    csd = make_csd(n, C)
    filt = onm.NxMFilter(t, csd, FS, pre, ignored_frequency_peaks=[60.0, 5000.0])
    tab = build_nxm_filter(t, csd, FS, pre, ignored_frequency_peaks=[60.0, 5000.0])
    k = n // 2 + 1
    assert tab.phi.shape == (M, C, k) and tab.icov.shape == (C, C, k) and tab.pinv.shape == (M, M)
    assert np.allclose(tab.phi, np.moveaxis(filt.phi, 0, 2)[:, :, :k], rtol=1e-12, atol=0)
    assert np.allclose(tab.icov, np.moveaxis(filt.icov, 0, 2)[:, :, :k], rtol=1e-12, atol=0)
    assert np.allclose(tab.pinv, filt.Pinv, rtol=1e-10)
    assert np.allclose(tab.ampres, filt.ampres, rtol=1e-10)
    assert nxm_search_range(n, pre, FS, -100, 100) == o1.search_range(filt, -100, 100)
    assert nxm_search_range(n, pre, FS, None, None, 10, 50) == o1.search_range(filt, None, None, 10, 50)
    with pytest.raises(ValueError):
        build_nxm_filter(t, csd[:, :, :-2], FS, pre)
    with pytest.raises(ValueError):
        build_nxm_filter(t[:1], csd, FS, pre)


# ------------------------------------------------------------------------------ GPU parity
def _check(plan, out, sid, ref, filt, nodelay=False):
    amps, t0, chi2, idx = plan.record(out.astype(np.float64), sid)
    key = "_nodelay" if nodelay else ""
    if not nodelay:
        want = ref["index"].astype(np.int64)
        got = idx.astype(np.int64)
        assert np.array_equal(got, want), f"time bins differ at {np.nonzero(got != want)[0]}"
        assert np.allclose(t0, ref["t0"], rtol=1e-6, atol=1e-12)
    else:
        assert np.all(idx == filt.pre) and np.all(t0 == 0)
    ra = ref["amps" + key]
    assert np.all(np.abs(amps - ra) <= AMP_RTOL * np.abs(ra) + AMP_ATOL_SIGMA * filt.ampres)
    rc = ref["chi2" + key]
    assert np.all(np.abs(chi2 - rc) <= CHI_RTOL * np.abs(rc) + CHI_ATOL_CHI0 * ref["chi2_0"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,pre,C,M", [(4096, 2048, 2, 2), (32768, 16384, 2, 2), (25000, 12500, 3, 2),
                                       (2048, 500, 1, 1), (4096, 2048, 4, 4), (8192, 4096, 2, 3),
                                       (250, 100, 2, 2), (1002, 400, 2, 1)])
def test_nxm_gpu_matches_oracle(n, pre, C, M):
    from detprocess_amd.ofnxm import NxMPlan, build_nxm_filter, nxm_search_range
    t = make_templates(n, pre, C, M)
    csd = make_csd(n, C)
    filt = onm.NxMFilter(t, csd, FS, pre)
    B = 24 if n <= 8192 else 10
    ev, _, _ = make_events(B, t, csd, filt.ampres, seed=n + C, max_delay=min(n // 8, 2000))
    plan = NxMPlan(build_nxm_filter(t, csd, FS, pre), max_batch=16)     # several chunks + a tail
    lo, hi = nxm_search_range(n, pre, FS, -100, 100)
    s_nd = plan.add_search("nodelay")
    s_un = plan.add_search("delay")
    s_co = plan.add_search("delay", lo, hi)
    s_out = plan.add_search("delay", lo, hi, outside=True)
    assert plan.row_floats == 4 * (M + 3)
    out = plan.process(ev.astype(np.float32))
    x = ev.astype(np.float32).astype(np.float64)
    r_un = onm.process_events(filt, x)
    _check(plan, out, s_nd, r_un, filt, nodelay=True)
    _check(plan, out, s_un, r_un, filt)
    _check(plan, out, s_co, onm.process_events(filt, x, window_min_from_trig_usec=-100,
                                               window_max_from_trig_usec=100), filt)
    _check(plan, out, s_out, onm.process_events(filt, x, window_min_from_trig_usec=-100,
                                                window_max_from_trig_usec=100,
                                                lgc_outside_window=True), filt)


@pytest.mark.gpu
def test_nxm_channel_map_valid_mask_and_device_buffers():
    import torch
    from detprocess_amd.ofnxm import NxMPlan, build_nxm_filter
    n, pre, C, M = 4096, 2048, 2, 2
    t = make_templates(n, pre, C, M)
    csd = make_csd(n, C)
    filt = onm.NxMFilter(t, csd, FS, pre)
    ev, _, _ = make_events(20, t, csd, filt.ampres, seed=9, max_delay=300)
    full = np.zeros((20, 4, n), dtype=np.float32)
    full[:, 3] = ev[:, 0]
    full[:, 1] = ev[:, 1]
    full[:, 0] = 1e-6                                         # channels the fit must not touch
    full[:, 2] = -1e-6
    plan = NxMPlan(build_nxm_filter(t, csd, FS, pre), max_batch=8)
    # first a call on the bare pair, then the map onto a wider event array (staging regrows)
    s0 = plan.add_search("delay", pre - 400, pre + 400)
    bare = plan.process(ev.astype(np.float32))
    plan.reset_searches()
    plan.set_channels(4, [3, 1])
    sid = plan.add_search("delay", pre - 400, pre + 400)
    valid = np.ones(20, dtype=np.uint8)
    valid[[2, 11]] = 0
    out = plan.process(full, valid)
    ref = onm.process_events(filt, ev.astype(np.float32).astype(np.float64),
                             window_min_index=pre - 400, window_max_index=pre + 400)
    ok = valid.astype(bool)
    assert np.all(out[~ok] == -999999.0)
    sub = {k: v[ok] for k, v in ref.items()}
    _check(plan, out[ok], sid, sub, filt)
    assert s0 == sid and np.array_equal(bare[ok], out[ok])
    # device-resident events: same numbers, no host staging
    dev = plan.process(torch.from_numpy(full).cuda(), torch.from_numpy(valid).cuda())
    assert np.array_equal(dev.cpu().numpy(), out)
    # an empty batch is legal and returns an empty matrix (host and device)
    assert plan.process(np.zeros((0, 4, n), dtype=np.float32)).shape == (0, plan.row_floats)
    assert tuple(plan.process(torch.zeros((0, 4, n), device="cuda")).shape) == (0, plan.row_floats)
    # a batch that is not a multiple of the workgroup's event group, nor of max_batch
    out13 = plan.process(full[:13], valid[:13])
    assert np.array_equal(out13, out[:13])
    # empty window -> sentinel record; bad shapes and missing searches raise
    plan.reset_searches()
    with pytest.raises(Exception):
        plan.process(full)
    s_empty = plan.add_search("delay", 100, 100)
    o2 = plan.process(full)
    assert np.all(plan.record(o2, s_empty)[2] == -999999.0)
    with pytest.raises(ValueError):
        plan.process(full[:, :2])


@pytest.mark.gpu
def test_nxm_1x1_agrees_with_the_of1x1_engine():
    """N = M = 1 is the single-template filter: the NxM engine and the of1x1 engines give the
    same bins and, within fp32 rounding, the same amplitudes and chi2."""
    from detprocess_amd import OFPlan, build_filter
    from detprocess_amd.ofnxm import NxMPlan, build_nxm_filter
    n, pre = 8192, 4096
    tmpl = synth.make_template(n, pre, FS)
    J = synth.make_psd(n, FS)
    ft = build_filter(tmpl, J, FS, pre)
    x, _, _ = synth.make_traces(32, tmpl, J, FS, ft.ampres, seed=4, max_delay=500)
    x = x.astype(np.float32)
    p1 = OFPlan(n, pre, FS, max_batch=64)
    p1.set_filter(0, ft)
    s1 = p1.add_search(0, "delay")
    o1x1 = p1.process(x)
    pn = NxMPlan(build_nxm_filter(tmpl[None, None], J[None, None], FS, pre), max_batch=64)
    sn = pn.add_search("delay")
    amps, t0, chi2, idx = pn.record(pn.process(x[:, None, :]), sn)
    off = p1.search_offset(0, s1)
    assert np.array_equal(idx, o1x1[:, off + 7])
    assert np.allclose(amps[:, 0], o1x1[:, off + 0], rtol=1e-4, atol=1e-3 * ft.ampres)
    assert np.allclose(chi2, o1x1[:, off + 2], rtol=1e-4)


@pytest.mark.gpu
def test_feature_extractors_ofnxm_static_method():
    """FeatureExtractors.ofnxm on an OFBase holding the per-channel signals, as
    ProcessingData drives it (processing_data.py:294-381, 746-772; algorithms.py:141-274)."""
    from detprocess_amd import FeatureExtractors as FE, OFBase
    n, pre, C, M = 4096, 2048, 2, 2
    t = make_templates(n, pre, C, M)
    csd = make_csd(n, C)
    filt = onm.NxMFilter(t, csd, FS, pre)
    ev, _, _ = make_events(6, t, csd, filt.ampres, seed=2, max_delay=100)
    ev32 = ev.astype(np.float32)
    ob = OFBase(FS)
    ob.set_csd("a|b", csd, coupling="AC")
    ob.add_template("a|b", t, template_tag="pair", pretrigger_samples=pre)
    ob.calc_phi("a|b", "pair")
    assert ob.phi("a|b", "pair").shape == (M, C, n // 2 + 1)
    assert np.allclose(np.sqrt(np.diag(ob.iweight("a|b", "pair"))), filt.ampres, rtol=1e-10)
    keys = ["chi2_ofnxm_constrained", "t0_ofnxm_constrained", "amp1_ofnxm_constrained",
            "amp2_ofnxm_constrained", "chi2_ofnxm_nodelay", "amp1_ofnxm_nodelay",
            "amp2_ofnxm_nodelay"]
    r = FE.ofnxm("a|b", ob, template_tag="pair")                     # no signal -> sentinels
    assert list(r) == keys and all(v == -999999.0 for v in r.values())
    with pytest.raises(ValueError):
        FE.ofnxm("a|b", ob)                                          # template tag required
    with pytest.raises(ValueError):
        FE.ofnxm("a|b", ob, template_tag="nope")
    with pytest.raises(ValueError):
        FE.ofnxm("a|b", ob, template_tag="pair", amplitude_names=["one"])
    ob.update_signal("a", ev32[:, 0], calc_fft=True)
    assert not ob.is_signal_stored("a|b")
    ob.update_signal("b", ev32[:, 1], calc_fft=True)
    assert ob.is_signal_stored("a|b")
    r = FE.ofnxm("a|b", ob, template_tag="pair", amplitude_names=["x", "y"],
                 window_min_from_trig_usec=-50, window_max_from_trig_usec=50,
                 feature_base_name="of2x2")
    ref = onm.process_events(filt, ev32.astype(np.float64), window_min_from_trig_usec=-50,
                             window_max_from_trig_usec=50)
    assert np.allclose(r["t0_of2x2_constrained"], ref["t0"], rtol=1e-6, atol=1e-12)
    for i, nm in enumerate(("x", "y")):
        assert np.allclose(r[f"{nm}_of2x2_constrained"], ref["amps"][:, i], rtol=AMP_RTOL,
                           atol=AMP_ATOL_SIGMA * filt.ampres[i])
        assert np.allclose(r[f"{nm}_of2x2_nodelay"], ref["amps_nodelay"][:, i], rtol=AMP_RTOL,
                           atol=AMP_ATOL_SIGMA * filt.ampres[i])
    assert np.allclose(r["chi2_of2x2_constrained"], ref["chi2"], rtol=CHI_RTOL,
                       atol=CHI_ATOL_CHI0 * ref["chi2_0"].max())
    assert np.allclose(r["chi2_of2x2_nodelay"], ref["chi2_nodelay"], rtol=CHI_RTOL,
                       atol=CHI_ATOL_CHI0 * ref["chi2_0"].max())
    # a single event comes back as scalars, as in the reference
    ob.clear_signal()
    ob.update_signal("a", ev32[0, 0])
    ob.update_signal("b", ev32[0, 1])
    r1 = FE.ofnxm("a|b", ob, template_tag="pair")
    assert isinstance(r1["amp1_ofnxm_constrained"], float)
    assert np.isclose(r1["amp1_ofnxm_nodelay"], ref["amps_nodelay"][0, 0], rtol=AMP_RTOL,
                      atol=AMP_ATOL_SIGMA * filt.ampres[0])


@pytest.mark.gpu
def test_nxm_on_events_cut_from_adc_streams():
    """NxM fit on events cut on the GPU out of int16 streams = the same fit on host-cut,
    host-converted events, bit for bit (plan level and through the YAML driver)."""
    import torch
    from detprocess_amd import FeatureProcessing, FilterData
    from detprocess_amd.ofnxm import NxMPlan, build_nxm_filter
    n, pre, C, M = 4096, 1024, 2, 2
    t = make_templates(n, pre, C, M)
    csd = make_csd(n, C)
    filt = onm.NxMFilter(t, csd, FS, pre)
    ev, _, _ = make_events(8, t, csd, filt.ampres, seed=5, max_delay=200)
    chans = ["x", "a", "b"]                                      # the pair sits in rows 1 and 2
    n_stream = 8 * n
    scale = np.array([1.0e-12, 2.0e-12, 2.5e-12])
    offset = np.array([0.0, -1e-9, 3e-10])
    adc = np.zeros((3, n_stream), dtype=np.int16)
    for c in range(2):
        train = ev[:, c].reshape(-1)
        adc[c + 1] = np.clip(np.round((train - offset[c + 1]) / scale[c + 1]), -32768, 32767)
    trig = np.array([pre, n + pre + 7, 3 * n + pre - 40, 7 * n + pre, 7 * n + pre + 1, 10],
                    dtype=np.int64)
    lo = trig - pre
    ok = (lo >= 0) & (lo + n <= n_stream)
    assert list(ok) == [True, True, True, True, False, False]
    cut = np.zeros((len(trig), 3, n), dtype=np.float32)
    for b in np.nonzero(ok)[0]:
        for c in range(3):
            cut[b, c] = (adc[c, lo[b]:lo[b] + n].astype(np.float32) * np.float32(scale[c])
                         + np.float32(offset[c]))
    plan = NxMPlan(build_nxm_filter(t, csd, FS, pre), max_batch=4)
    plan.set_channels(3, [1, 2])
    plan.add_search("nodelay")
    plan.add_search("delay", pre - 300, pre + 300)
    want = plan.process(cut, ok.astype(np.uint8))
    got = plan.process_adc(adc, trig, scale, offset)
    assert np.array_equal(got, want)
    assert np.all(got[~ok] == -999999.0)
    got_dev = plan.process_adc(torch.from_numpy(adc).cuda(), trig, scale, offset)
    assert np.array_equal(got_dev.cpu().numpy(), want)
    # YAML driver: an a|b block next to a single-channel block, both from the same streams
    fd = FilterData()
    f = np.fft.fftfreq(n, d=1 / FS)
    fd.set_template("a|b", t, sample_rate=FS, pretrigger_length_samples=pre, tag="pair")
    fd.set_csd("a|b", csd, f, sample_rate=FS, tag="default")
    fd.set_template("a", t[0, 0] / np.max(np.abs(t[0, 0])), sample_rate=FS,
                    pretrigger_length_samples=pre, tag="default")
    fd.set_psd("a", csd[0, 0].real, f, sample_rate=FS, tag="default")
    yaml_text = """
a:
    of1x1_nodelay:
        run: True
        template_tag: default
a|b:
    feature_channel: ab
    of2x2:
        run: True
        base_algorithm: ofnxm
        template_tag: pair
        window_min_from_trig_usec: -200
        window_max_from_trig_usec: 200
"""
    fp = FeatureProcessing(yaml_text, fd, chans, FS, nb_samples=n, nb_pretrigger_samples=pre)
    df = fp.process_adc(adc, trig, scale, offset)
    df2 = fp.process(cut, valid=ok.astype(np.uint8))
    assert list(df.columns) == list(df2.columns) and "amp1_of2x2_constrained_ab" in df.columns
    assert np.array_equal(df.to_numpy(), df2.to_numpy())
    assert (df.iloc[4] == -999999.0).all()


@pytest.mark.gpu
def test_nxm_properties_at_full_size():
    """Size-independent properties on a batch the oracle could not check in seconds (2048 events,
    2 x 2, 32768 samples, device-resident): exact scaling by powers of two, window consistency,
    chi2 ordering, circular-shift equivariance."""
    import torch
    from detprocess_amd.ofnxm import NxMPlan, build_nxm_filter
    n, pre, C, M, B = 32768, 16384, 2, 2, 2048
    t = make_templates(n, pre, C, M)
    csd = make_csd(n, C)
    tab = build_nxm_filter(t, csd, FS, pre)
    g = torch.Generator(device="cuda").manual_seed(3)
    ev = torch.randn((B, C, n), device="cuda", generator=g) * 2e-9
    tt = torch.as_tensor(t, dtype=torch.float32, device="cuda")
    amp = 1e-7 + torch.rand((B, M), device="cuda", generator=g) * 3e-7     # well above the noise
    shift = torch.randint(-300, 301, (B,), device="cuda", generator=g)
    pulses = torch.einsum("bm,amn->ban", amp, tt)
    idx = (torch.arange(n, device="cuda")[None, :] - shift[:, None]) % n
    ev += torch.gather(pulses, 2, idx[:, None, :].expand(B, C, n))
    plan = NxMPlan(tab, max_batch=512)
    s_nd = plan.add_search("nodelay")
    s_un = plan.add_search("delay")
    s_win = plan.add_search("delay", pre - 400, pre + 400)
    out = plan.process(ev)
    a_nd, _, c_nd, _ = plan.record(out, s_nd)
    a_un, t_un, c_un, i_un = plan.record(out, s_un)
    a_w, t_w, c_w, i_w = plan.record(out, s_win)
    # the injected delay is found (SNR is high) and the window that contains it changes nothing
    assert torch.equal(i_un.long(), (pre + shift).long())
    assert torch.equal(i_w, i_un) and torch.equal(a_w, a_un) and torch.equal(c_w, c_un)
    assert torch.allclose(a_un, amp, rtol=0.05, atol=2e-8)
    # a delay fit can only lower chi2; at zero delay both are the same fit
    assert bool((c_un <= c_nd * (1 + 1e-6)).all())
    zero = shift == 0
    if bool(zero.any()):
        assert torch.allclose(c_un[zero], c_nd[zero], rtol=1e-6)
    # scaling the events by 4 scales amplitudes by 4 and chi2 by 16, bit for bit
    out4 = plan.process(ev * 4.0)
    a4, t4, c4, i4 = plan.record(out4, s_un)
    assert torch.equal(i4, i_un) and torch.equal(a4, a_un * 4.0) and torch.equal(c4, c_un * 16.0)
    # rolling every channel by d samples moves the time bin by d and leaves the fit unchanged
    d = 37
    outr = plan.process(torch.roll(ev, d, dims=2))
    ar, tr, cr, ir = plan.record(outr, s_un)
    assert torch.equal(ir.long(), i_un.long() + d)
    assert torch.allclose(ar, a_un, rtol=2e-5, atol=1e-4 * float(tab.ampres.max()))
    assert torch.allclose(cr, c_un, rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n,pre,C,M", [(4096, 2048, 2, 2), (32768, 16384, 3, 2)])
def test_nxm_interpolate_t0(n, pre, C, M):
    """interpolate_t0 (algorithms.py:152, 259): the refined delay fit against the oracle's
    restatement, on pulses clear of the noise (the sub-sample offset is a ratio of differences
    whose fp32 error scales as 1 / SNR, as for of1x1); the time bin itself stays exact, and
    the unrefined search of the same plan is untouched."""
    from detprocess_amd.ofnxm import NxMPlan, build_nxm_filter, nxm_search_range
    t = make_templates(n, pre, C, M)
    csd = make_csd(n, C)
    filt = onm.NxMFilter(t, csd, FS, pre)
    ev, _, _ = make_events(32, t, csd, filt.ampres, seed=3, max_delay=min(n // 8, 2000))
    plan = NxMPlan(build_nxm_filter(t, csd, FS, pre), max_batch=16)
    lo, hi = nxm_search_range(n, pre, FS, -400, 400)
    s_plain = plan.add_search("delay", lo, hi)
    s_int = plan.add_search("delay", lo, hi, interpolate=True)
    s_full = plan.add_search("delay", interpolate=True)
    x32 = ev.astype(np.float32)
    out = plan.process(x32).astype(np.float64)
    x = x32.astype(np.float64)
    _check(plan, out, s_plain, onm.process_events(filt, x, window_min_from_trig_usec=-400,
                                                  window_max_from_trig_usec=400), filt)
    for sid, kw in ((s_int, dict(window_min_from_trig_usec=-400, window_max_from_trig_usec=400)),
                    (s_full, {})):
        ref = onm.process_events(filt, x, interpolate_t0=True, **kw)
        amps, t0, chi2, idx = plan.record(out, sid)
        assert np.array_equal(idx.astype(np.int64), ref["index"])
        snr = np.sqrt(np.maximum(ref["chi2_0"] - ref["chi2"], 0.0))
        hi_snr = snr > 50
        assert hi_snr.sum() >= 5
        assert np.all(np.abs(t0 - ref["t0"])[hi_snr] <= 2e-3 / FS)
        assert np.all(np.abs(t0 * FS - (idx - pre)) <= 1.0 + 1e-6)          # within one bin
        ra = ref["amps"]
        assert np.all((np.abs(amps - ra) <= 3e-5 * np.abs(ra) + 2e-4 * filt.ampres)[hi_snr])
        assert np.all((np.abs(chi2 - ref["chi2"]) <= CHI_RTOL * np.abs(ref["chi2"])
                       + 4e-6 * ref["chi2_0"])[hi_snr])
    # refined and unrefined differ (the refinement does something) but by less than a bin
    _, t0p, _, _ = plan.record(out, s_plain)
    _, t0i, _, _ = plan.record(out, s_int)
    assert np.any(t0p != t0i) and np.all(np.abs(t0p - t0i) <= 1.0 / FS + 1e-12)
