"""GPU parity tests of the one-wave-per-trace kernel for 4096-sample traces (BASELINE configs[0]'s
trace length; detprocess_amd/csrc/ofx_wave.hip): k_wave through the C ABI against the fp64 oracle,
and the LDS engine on the same inputs (the golden fixtures of this length: tests/test_golden.py).
Every test also runs at 8192 and 16384 samples: k_wave2 (ofx_wave2.hip) with two / four waves per trace."""
import numpy as np
import pytest

from detprocess_amd import build_filter, synth
from oracle import of1x1 as orc
from util import check_search, combine_fp32

pytestmark = pytest.mark.gpu
FS = 1.25e6
N = 4096


@pytest.fixture(autouse=True, params=[4096, 8192, 16384])
def _trace_length(request):
    global N
    N = request.param
    yield
    N = 4096


def _mk(pre=None, engine="fused", max_batch=64):
    from detprocess_amd import OFPlan
    pre = N // 2 if pre is None else pre
    tmpl = synth.make_template(N, pre, FS)
    psd = synth.make_psd(N, FS)
    ft = build_filter(tmpl, psd, FS, pre)
    filt = orc.OFFilter(tmpl, psd, FS, pre)
    plan = OFPlan(N, pre, FS, max_batch=max_batch, device=0, engine=engine)
    plan.set_filter(0, ft)
    return plan, ft, filt, tmpl, psd


def _run(plan, x32, **kw):
    import torch
    return plan.process(torch.as_tensor(x32, device="cuda:0"), **kw).cpu().numpy().astype(np.float64)


def test_auto_picks_the_wave_kernel_at_4096():
    plan, *_ = _mk(engine="auto")
    assert plan.engine == "fused"


@pytest.mark.parametrize("B", [1, 3, 37, 4100])
def test_unconstrained_vs_oracle(B):
    plan, ft, filt, tmpl, psd = _mk(max_batch=8192)
    sid = plan.add_search(0, "delay")
    x, _, _ = synth.make_traces(B, tmpl, psd, FS, ft.ampres, seed=100 + B, max_delay=N // 8)
    x32 = x.astype(np.float32)
    out = _run(plan, x32)
    ref = orc.process_events(filt, x32.astype(np.float64), "unconstrained")
    check_search(out, plan.search_offset(0, sid), ref, "", ft.ampres, FS, "4096/wave")


def test_every_lag_wins():
    """A noiseless template shifted to EVERY lag of the trace: every lane, register and component of
    the lag dump, both ends of the rolled range."""
    plan, ft, filt, tmpl, psd = _mk(max_batch=8192)
    sid = plan.add_search(0, "delay")
    lags = np.arange(-N // 2, N // 2)
    x = np.stack([3e-7 * np.roll(tmpl, int(d)) for d in lags]).astype(np.float32)
    out = _run(plan, x)
    o = plan.search_offset(0, sid)
    assert np.array_equal(out[:, o + 7].astype(int), N // 2 + lags)
    assert np.allclose(out[:, o + 0], 3e-7, rtol=2e-6)


def test_every_bin_passes_the_middle_step():
    """One cosine per one-sided bin k (the round-3 probe of k_fused25, tools/probe_bins.py): the
    amplitude at lag 0 and chi2_0 of a single-bin trace follow from the filter in closed form, so a
    bin that lands in the wrong slot of the pairwise middle step -- lane 0's permuted blocks, the
    self-paired M/2, DC / Nyquist -- shows up by name."""
    plan, ft, filt, tmpl, psd = _mk(max_batch=4096)
    sid = plan.add_search(0, "nodelay")
    K = N // 2 + 1
    t = np.arange(N)
    ks = np.arange(K)
    x = np.stack([1e-7 * np.cos(2 * np.pi * k * t / N + 0.3) for k in ks]).astype(np.float32)
    out = _run(plan, x)
    ref = orc.process_events(filt, x.astype(np.float64), "nodelay")
    o = plan.search_offset(0, sid)
    sc_a = np.abs(ref["amp"]).max()
    bad_a = np.nonzero(np.abs(out[:, o + 0] - ref["amp"]) > 2e-5 * np.abs(ref["amp"]) + 1e-5 * sc_a)[0]
    assert len(bad_a) == 0, f"amplitude wrong at bins {bad_a[:20]}"
    bad_c = np.nonzero(np.abs(out[:, o + 4] - ref["chi2nopulse"]) > 3e-5 * ref["chi2nopulse"] + 1e-30)[0]
    assert len(bad_c) == 0, f"chi2_0 wrong at bins {bad_c[:20]}"


def test_interpolate_windows_and_cutoffs():
    plan, ft, filt, tmpl, psd = _mk()
    pre = N // 2
    s1 = plan.add_search(0, "delay", interpolate=True)
    s2 = plan.add_search(0, "delay", pre - 300, pre + 300, interpolate=True, lowchi2_fcutoff=19000.0)
    s3 = plan.add_search(0, "delay", lowchi2_fcutoff=50000.0)
    s4 = plan.add_search(0, "delay", 0, 1, interpolate=True)
    s5 = plan.add_search(0, "nodelay", lowchi2_fcutoff=77000.0)
    s6 = plan.add_search(0, "delay", pre - 500, pre + 500, outside=True)
    s7 = plan.add_search(0, "delay", 0, 300)
    x, _, _ = synth.make_traces(41, tmpl, psd, FS, ft.ampres, seed=77, max_delay=N // 16)
    x32 = x.astype(np.float32)
    x64 = x32.astype(np.float64)
    out = _run(plan, x32)
    assert plan.engine == "fused"
    P = lambda *a, **k: orc.process_events(filt, x64, *a, **k)
    check_search(out, plan.search_offset(0, s1), P("unconstrained", interpolate=True), "", ft.ampres, FS,
                 "interp", interpolated=True)
    check_search(out, plan.search_offset(0, s2),
                 P("constrained", lowchi2_fcutoff=19000.0, interpolate=True, window_min_index=pre - 300,
                   window_max_index=pre + 300), "", ft.ampres, FS, "interp-win", interpolated=True,
                 lowchi2_fcutoff=19000.0)
    check_search(out, plan.search_offset(0, s3), P("unconstrained", lowchi2_fcutoff=50000.0), "", ft.ampres,
                 FS, "50 kHz", lowchi2_fcutoff=50000.0)
    check_search(out, plan.search_offset(0, s4),
                 P("constrained", interpolate=True, window_min_index=0, window_max_index=1), "", ft.ampres,
                 FS, "interp-edge", interpolated=True)
    check_search(out, plan.search_offset(0, s5), P("nodelay", lowchi2_fcutoff=77000.0), "", ft.ampres, FS,
                 "nodelay 77 kHz", lowchi2_fcutoff=77000.0)
    check_search(out, plan.search_offset(0, s6),
                 P("constrained", window_min_index=pre - 500, window_max_index=pre + 500,
                   lgc_outside_window=True), "", ft.ampres, FS, "outside")
    check_search(out, plan.search_offset(0, s7), P("constrained", window_min_index=0, window_max_index=300),
                 "", ft.ampres, FS, "early window")


@pytest.mark.parametrize("where", [0, 1, 2, 3])
def test_pretrigger_away_from_the_middle(where):
    pre = [7, 1000, (4 * N) // 5 + 1, N - 10][where]
    plan, ft, filt, tmpl, psd = _mk(pre=pre)
    s0 = plan.add_search(0, "nodelay")
    s1 = plan.add_search(0, "delay")
    lo, hi = max(0, pre - 200), min(N, pre + 300)
    s2 = plan.add_search(0, "delay", lo, hi, interpolate=True)
    s3 = plan.add_search(0, "delay", lo, hi, outside=True)
    x, _, _ = synth.make_traces(13, tmpl, psd, FS, ft.ampres, seed=pre, max_delay=150)
    x32 = x.astype(np.float32)
    x64 = x32.astype(np.float64)
    out = _run(plan, x32)
    check_search(out, plan.search_offset(0, s0), orc.process_events(filt, x64, "nodelay"), "", ft.ampres, FS,
                 f"pre{pre}/nodelay")
    check_search(out, plan.search_offset(0, s1), orc.process_events(filt, x64, "unconstrained"), "", ft.ampres,
                 FS, f"pre{pre}/delay")
    check_search(out, plan.search_offset(0, s2),
                 orc.process_events(filt, x64, "constrained", interpolate=True, window_min_index=lo,
                                    window_max_index=hi), "", ft.ampres, FS, f"pre{pre}/window",
                 interpolated=True)
    check_search(out, plan.search_offset(0, s3),
                 orc.process_events(filt, x64, "constrained", window_min_index=lo, window_max_index=hi,
                                    lgc_outside_window=True), "", ft.ampres, FS, f"pre{pre}/outside")


def test_windows_bands_and_channel_algebra():
    import torch
    from detprocess_amd import OFPlan
    pre = N // 2 - 11
    psd = synth.make_psd(N, FS)
    tmpl = synth.make_template(N, pre, FS)
    ft = build_filter(tmpl, psd, FS, pre)
    filt = orc.OFFilter(tmpl, psd, FS, pre)
    plan = OFPlan(N, pre, FS, max_batch=64, device=0, engine="fused")
    plan.set_filter(0, ft)
    ids = (plan.add_search(0, "nodelay", lowchi2_fcutoff=50000.0), plan.add_search(0, "delay", lowchi2_fcutoff=50000.0),
           plan.add_search(0, "delay", pre - 500, pre + 500, interpolate=True))
    wins = [(100, 1500), (0, N - 1), (N // 2 - 500, N // 2 + 263), (256, 512), (255, 513), (1, 2), (4095, 4096),
            (0, N), (N // 2 - 1, N // 2 + 1), (N - 1, N)][:8 if N == 4096 else 10][-8:]
    wid = [plan.add_tdwindow(a, b) for a, b in wins]
    bands = [(1, 20), (N // 60, N // 36), (200, 256)]
    bid = [plan.add_band(a, b) for a, b in bands]
    ev, _, _ = synth.make_traces(2 * 21, tmpl, psd, FS, ft.ampres, seed=314)
    ev = ev.reshape(21, 2, N).astype(np.float32)
    plan.set_channels(2, [0, 1], [0.75, -1.25])
    out = plan.process(torch.as_tensor(ev, device="cuda:0")).cpu().numpy().astype(np.float64)
    assert plan.engine == "fused"
    x64 = combine_fp32(ev, [0, 1], [0.75, -1.25])
    r_nd = orc.process_events(filt, x64, "nodelay", lowchi2_fcutoff=50000.0)
    r_un = orc.process_events(filt, x64, "unconstrained", lowchi2_fcutoff=50000.0)
    r_co = orc.process_events(filt, x64, "constrained", window_min_index=pre - 500, window_max_index=pre + 500,
                              interpolate=True)
    check_search(out, plan.search_offset(0, ids[0]), r_nd, "", ft.ampres, FS, "nodelay", lowchi2_fcutoff=50000.0)
    check_search(out, plan.search_offset(0, ids[1]), r_un, "", ft.ampres, FS, "delay", lowchi2_fcutoff=50000.0)
    check_search(out, plan.search_offset(0, ids[2]), r_co, "", ft.ampres, FS, "window", interpolated=True)
    sc = np.abs(x64).max()
    x32r = x64.astype(np.float32).astype(np.float64)
    for i, (a, b) in enumerate(wins):
        t = plan.tdwindow_offset(wid[i])
        assert np.allclose(out[:, t + 0], orc.baseline(x64, a, b), rtol=1e-4, atol=1e-6 * sc), (a, b)
        assert np.allclose(out[:, t + 1], orc.integral(x64, FS, a, b), rtol=1e-4, atol=1e-6 * sc * (b - a) / FS), (a, b)
        assert np.array_equal(out[:, t + 2], x32r[:, a:b].max(axis=1)), (a, b)
        assert np.array_equal(out[:, t + 3], x32r[:, a:b].min(axis=1)), (a, b)
    for i, (lo, hi) in enumerate(bands):
        V = np.fft.rfft(x64, axis=-1)[:, lo:hi] / N
        want = np.sqrt(2.0 * np.abs(V) ** 2 * N / FS).mean(axis=-1)
        assert np.allclose(out[:, plan.band_offset(bid[i])], want, rtol=2e-5), (lo, hi)
    # windows only (no slot with a search): the kernel stops after the window sums
    pw = OFPlan(N, pre, FS, max_batch=64, device=0, engine="fused")
    w2 = [pw.add_tdwindow(a, b) for a, b in wins]
    ow = pw.process(torch.as_tensor(ev[:, 0].copy(), device="cuda:0")).cpu().numpy().astype(np.float64)
    x1 = ev[:, 0].astype(np.float64)
    for i, (a, b) in enumerate(wins):
        t = pw.tdwindow_offset(w2[i])
        assert np.allclose(ow[:, t + 0], orc.baseline(x1, a, b), rtol=1e-4, atol=1e-6 * sc), (a, b)
        assert np.array_equal(ow[:, t + 2], x1[:, a:b].max(axis=1)), (a, b)


def test_edge_cases_and_fallback():
    import torch
    from detprocess_amd import _lib
    plan, ft, filt, tmpl, psd = _mk()
    sid = plan.add_search(0, "delay")
    plan.add_search(0, "nodelay")
    wid = plan.add_tdwindow(0, N - 1)
    x = np.zeros((6, N), dtype=np.float32)
    x[1] = 3e-8
    x[2, 1234] = 1e-7
    x[3] = (2e-7 * np.roll(tmpl, -400)).astype(np.float32)
    x[4] = x[3]
    x[5] = x[3]
    x[5, 777] = np.nan
    out = _run(plan, x)
    o = plan.search_offset(0, sid)
    assert out[0, o + 7] == 0 and out[0, o + 0] == 0 and out[0, o + 2] == 0     # all-zero: first rolled bin
    ref = orc.process_events(filt, x[:5].astype(np.float64), "unconstrained")
    assert np.array_equal(out[2:5, o + 7].astype(int), ref["index"][2:])
    assert abs(out[1, o + 0]) < 1e-3 * ft.ampres
    assert out[3, o + 7] == N // 2 - 400
    assert np.allclose(out[2:5, o + 0], ref["amp"][2:], rtol=1e-5, atol=1e-4 * ft.ampres)
    assert np.isnan(out[5, o + 0]) and np.isnan(out[5, o + 2])                   # NaN trace -> NaN record
    t = plan.tdwindow_offset(wid)
    assert np.allclose(out[:5, t + 0], x[:5, :N - 1].astype(np.float64).mean(axis=1), rtol=1e-5, atol=1e-13)
    valid = torch.tensor([1, 0, 1, 0, 1, 0], dtype=torch.uint8, device="cuda:0")
    out2 = plan.process(torch.as_tensor(x, device="cuda:0"), valid=valid).cpu().numpy()
    assert np.all(out2[[1, 3, 5]] == -999999.0)
    assert np.array_equal(out2[[0, 2, 4]].astype(np.float64), out[[0, 2, 4]])
    e = plan.process(torch.empty((0, N), dtype=torch.float32, device="cuda:0"))
    assert tuple(e.shape) == (0, plan.row_floats)
    # beyond the 256 stashed bins: AUTO falls back for that call (LDS engine), FUSED refuses
    xs, _, _ = synth.make_traces(9, tmpl, psd, FS, ft.ampres, seed=3)
    x32 = xs.astype(np.float32)
    plan_a, ft, filt, tmpl, psd = _mk(engine="auto")
    sa = plan_a.add_search(0, "delay", lowchi2_fcutoff=100000.0)
    outa = _run(plan_a, x32)
    ref = orc.process_events(filt, x32.astype(np.float64), "unconstrained", lowchi2_fcutoff=100000.0)
    check_search(outa, plan_a.search_offset(0, sa), ref, "", ft.ampres, FS, "auto/100kHz", lowchi2_fcutoff=100000.0)
    plan_f, *_ = _mk(engine="fused")
    plan_f.add_search(0, "delay", lowchi2_fcutoff=100000.0)
    with pytest.raises(_lib.OfxError):
        _run(plan_f, x32)
    # several filter slots: one launch per slot
    plan_m, ft, filt, tmpl, psd = _mk(engine="fused")
    t2 = synth.make_template(N, N // 2, FS, "glitch")
    ft2 = build_filter(t2, psd, FS, N // 2)
    plan_m.set_filter(1, ft2)
    m0, m1 = plan_m.add_search(0, "delay"), plan_m.add_search(1, "delay")
    wm = plan_m.add_tdwindow(10, 900)
    outm = _run(plan_m, x32)
    assert plan_m.engine == "fused"
    assert np.allclose(outm[:, plan_m.tdwindow_offset(wm)], x32[:, 10:900].astype(np.float64).mean(axis=1),
                       rtol=1e-4, atol=1e-13)
    check_search(outm, plan_m.search_offset(0, m0), orc.process_events(filt, x32.astype(np.float64), "unconstrained"),
                 "", ft.ampres, FS, "two slots 0")
    check_search(outm, plan_m.search_offset(1, m1),
                 orc.process_events(orc.OFFilter(t2, psd, FS, N // 2), x32.astype(np.float64), "unconstrained"),
                 "", ft2.ampres, FS, "two slots 1")


def test_agrees_with_the_lds_engine_at_scale():
    """65536 generated traces: same t0 bins as the LDS engine, amplitudes and chi2 within the fp32
    error of either engine; the same launch twice is bit-identical; linearity and shift."""
    import torch
    plan, ft, filt, tmpl, psd = _mk(max_batch=65536)
    pl2, *_ = _mk(engine="lds", max_batch=65536)
    plan.add_search(0, "delay"); pl2.add_search(0, "delay")
    from detprocess_amd import synth_traces
    B = 65536
    x, _ = synth_traces(B, N, tmpl, 0.0, 30 * ft.ampres, 300 * ft.ampres, 0.5, N // 16, seed=5, psd=psd, fs=FS)
    a_t = plan.process(x)
    a = a_t.cpu().numpy().astype(np.float64)
    b = pl2.process(x).cpu().numpy().astype(np.float64)
    flips = np.nonzero(a[:, 7] != b[:, 7])[0]
    assert len(flips) <= 4, f"{len(flips)} t0 bins differ"
    keep = np.ones(B, bool); keep[flips] = False
    assert np.all(np.abs(a[keep, 0] - b[keep, 0]) <= 2e-5 * np.abs(b[keep, 0]) + 2e-4 * ft.ampres)
    assert np.all(np.abs(a[keep, 2] - b[keep, 2]) <= 2e-5 * np.abs(b[keep, 2]) + 4e-6 * b[keep, 4])
    assert np.allclose(a[:, 4], b[:, 4], rtol=1e-5)
    assert torch.equal(plan.process(x), a_t)
    a2 = plan.process(x[:4096] * 2.0).cpu().numpy().astype(np.float64)
    same = a2[:, 7] == a[:4096, 7]
    assert same.mean() > 0.999
    assert np.allclose(a2[same, 0], 2.0 * a[:4096][same, 0], rtol=1e-5, atol=1e-4 * ft.ampres)


@pytest.mark.parametrize("feat", range(8))
def test_every_kernel_instantiation_vs_oracle(feat):
    import torch
    from detprocess_amd import OFPlan
    pre, B = N // 2 + 37, 48
    psd = synth.make_psd(N, FS)
    tmpl = synth.make_template(N, pre, FS)
    ft = build_filter(tmpl, psd, FS, pre)
    filt = orc.OFFilter(tmpl, psd, FS, pre)
    plan = OFPlan(N, pre, FS, max_batch=64, device=0, engine="fused")
    plan.set_filter(0, ft)
    ids = [("nodelay", plan.add_search(0, "nodelay")), ("unconstrained", plan.add_search(0, "delay"))]
    if feat & 1:
        ids.append(("constrained", plan.add_search(0, "delay", pre - 400, pre + 400)))
    wins = [(N // 10, N // 2), (N // 2 - 300, N // 2 + 900)] if feat & 2 else []
    wid = [plan.add_tdwindow(a, b) for a, b in wins]
    nch = 2 if feat & 4 else 1
    ev, _, _ = synth.make_traces(B * nch, tmpl, psd, FS, ft.ampres, seed=5 + feat, max_delay=N // 16)
    ev = ev.reshape(B, nch, N).astype(np.float32)
    if feat & 4:
        plan.set_channels(2, [1, 0], [1.0, -0.5])
        x64 = combine_fp32(ev, [1, 0], [1.0, -0.5])
        out = plan.process(torch.as_tensor(ev, device="cuda:0")).cpu().numpy().astype(np.float64)
    else:
        x64 = ev[:, 0].astype(np.float64)
        out = plan.process(torch.as_tensor(ev[:, 0], device="cuda:0")).cpu().numpy().astype(np.float64)
    assert plan.engine == "fused"
    for mode, sid in ids:
        kw = dict(window_min_index=pre - 400, window_max_index=pre + 400) if mode == "constrained" else {}
        r = orc.process_events(filt, x64, mode, **kw)
        check_search(out, plan.search_offset(0, sid), r, "", ft.ampres, FS, f"k_wave<{feat}> {mode}")
    sc = np.abs(x64).max()
    for (a, b), w in zip(wins, wid):
        t_ = plan.tdwindow_offset(w)
        assert np.allclose(out[:, t_ + 0], orc.baseline(x64, a, b), rtol=1e-4, atol=1e-6 * sc)
        assert np.allclose(out[:, t_ + 2], x64[:, a:b].max(axis=1), rtol=2e-6, atol=1e-7 * sc)
        assert np.allclose(out[:, t_ + 3], x64[:, a:b].min(axis=1), rtol=2e-6, atol=1e-7 * sc)


def test_three_slots_windows_bands_and_channel_algebra():
    """The example YAML's shape: three template tags on one plan (one launch per slot here), windows,
    bands, a summed channel -- against the oracle, every slot bit-identical to a single-slot plan."""
    import torch
    from detprocess_amd import OFPlan
    pre = N // 2
    psd = synth.make_psd(N, FS)
    kinds = ("pulse", "glitch", "muon")
    tmpls = [synth.make_template(N, pre, FS, k) for k in kinds]
    fts = [build_filter(t, psd, FS, pre) for t in tmpls]
    filts = [orc.OFFilter(t, psd, FS, pre) for t in tmpls]
    plan = OFPlan(N, pre, FS, max_batch=64, device=0, engine="fused")
    ids = []
    for s, ft in enumerate(fts):
        plan.set_filter(s, ft)
        ids.append((plan.add_search(s, "nodelay", lowchi2_fcutoff=50000.0),
                    plan.add_search(s, "delay", lowchi2_fcutoff=50000.0),
                    plan.add_search(s, "delay", pre - 500, pre + 500, interpolate=(s == 1))))
    wins = [(100, 1500), (0, N - 1), (N // 2 - 500, N // 2 + 263)]
    wid = [plan.add_tdwindow(a, b) for a, b in wins]
    bands = [(1, 20), (N // 60, N // 36), (160, 200)]
    bid = [plan.add_band(a, b) for a, b in bands]
    ev, _, _ = synth.make_traces(2 * 21, tmpls[0], psd, FS, fts[0].ampres, seed=314)
    ev = ev.reshape(21, 2, N).astype(np.float32)
    plan.set_channels(2, [0, 1], [0.75, -1.25])
    out = plan.process(torch.as_tensor(ev, device="cuda:0")).cpu().numpy().astype(np.float64)
    assert plan.engine == "fused"
    x64 = combine_fp32(ev, [0, 1], [0.75, -1.25])
    for s, (ft, filt) in enumerate(zip(fts, filts)):
        r_nd = orc.process_events(filt, x64, "nodelay", lowchi2_fcutoff=50000.0)
        r_un = orc.process_events(filt, x64, "unconstrained", lowchi2_fcutoff=50000.0)
        r_co = orc.process_events(filt, x64, "constrained", window_min_index=pre - 500,
                                  window_max_index=pre + 500, interpolate=(s == 1))
        for j, r in enumerate((r_nd, r_un, r_co)):
            check_search(out, plan.search_offset(s, ids[s][j]), r, "", ft.ampres, FS, f"{kinds[s]} search {j}",
                         interpolated=(j == 2 and s == 1), lowchi2_fcutoff=50000.0 if j < 2 else 10000.0)
    sc = np.abs(x64).max()
    for i, (a, b) in enumerate(wins):
        t = plan.tdwindow_offset(wid[i])
        assert np.allclose(out[:, t + 0], orc.baseline(x64, a, b), rtol=1e-4, atol=1e-6 * sc), (a, b)
        assert np.allclose(out[:, t + 1], orc.integral(x64, FS, a, b), rtol=1e-4, atol=1e-6 * sc * (b - a) / FS), (a, b)
    for i, (lo, hi) in enumerate(bands):
        V = np.fft.rfft(x64, axis=-1)[:, lo:hi] / N
        want = np.sqrt(2.0 * np.abs(V) ** 2 * N / FS).mean(axis=-1)
        assert np.allclose(out[:, plan.band_offset(bid[i])], want, rtol=2e-5), (lo, hi)
    for s, ft in enumerate(fts):
        solo = OFPlan(N, pre, FS, max_batch=64, device=0, engine="fused")
        solo.set_filter(0, ft)
        solo.add_search(0, "nodelay", lowchi2_fcutoff=50000.0)
        solo.add_search(0, "delay", lowchi2_fcutoff=50000.0)
        solo.add_search(0, "delay", pre - 500, pre + 500, interpolate=(s == 1))
        solo.set_channels(2, [0, 1], [0.75, -1.25])
        so = solo.process(torch.as_tensor(ev, device="cuda:0")).cpu().numpy().astype(np.float64)
        o0 = plan.search_offset(s, ids[s][0])
        assert np.array_equal(out[:, o0:o0 + 24], so[:, :24]), f"slot {s} differs from its solo plan"
