"""GPU parity tests of the register-resident kernel for 25000-sample traces (the trace length of
the reference's example YAML, examples/processing/process_example.yaml:93): k_fused25 through the
C ABI against the fp64 oracle, and against the LDS engine on the same inputs.  Every test also
runs at 12500 samples (10 ms traces, examples/filterdata/filter_data_generation.ipynb: the same
source built with a 10-point first stage, k_fused12)."""
import numpy as np
import pytest

from detprocess_amd import build_filter, synth
from oracle import of1x1 as orc
from util import check_search, check_td, combine_fp32

pytestmark = pytest.mark.gpu
FS = 1.25e6
N = 25000


@pytest.fixture(autouse=True, params=[25000, 12500, 20000])
def _trace_length(request):
    global N
    N = request.param
    yield
    N = 25000


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


def test_auto_picks_the_fused_kernel_at_25000():
    plan, *_ = _mk(engine="auto")
    assert plan.engine == "fused"


@pytest.mark.parametrize("B", [1, 37, 1100])
def test_unconstrained_vs_oracle(B):
    plan, ft, filt, tmpl, psd = _mk()
    sid = plan.add_search(0, "delay")
    x, _, _ = synth.make_traces(B, tmpl, psd, FS, ft.ampres, seed=100 + B, max_delay=N // 8)
    x32 = x.astype(np.float32)
    out = _run(plan, x32)
    ref = orc.process_events(filt, x32.astype(np.float64), "unconstrained")
    check_search(out, plan.search_offset(0, sid), ref, "", ft.ampres, FS, "25000/fused")


def test_every_lag_can_win():
    """A noiseless template shifted to lags spread over the whole trace (every register row,
    both components, all three rounds of virtual threads, both ends of the rolled range)."""
    plan, ft, filt, tmpl, psd = _mk()
    sid = plan.add_search(0, "delay")
    H, R = N // 2, N // 1250
    lags = np.unique(np.concatenate([np.arange(-H, H, 311), [-H, -H + 1, -1, 0, 1, H - 2, H - 1],
                                     1250 * np.arange(-R // 2, R // 2), 1250 * np.arange(-R // 2, R // 2) + 1249,
                                     [498, 499, 500, 501, 998, 999, 1000, 1001, 248, 249, 250, 251, 123, 124, 125, 126]]))
    x = np.stack([3e-7 * np.roll(tmpl, int(d)) for d in lags]).astype(np.float32)
    out = _run(plan, x)
    o = plan.search_offset(0, sid)
    assert np.array_equal(out[:, o + 7].astype(int), N // 2 + lags)
    assert np.allclose(out[:, o + 0], 3e-7, rtol=2e-6)


def test_interpolate_windows_and_cutoffs():
    plan, ft, filt, tmpl, psd = _mk()
    pre = N // 2
    s1 = plan.add_search(0, "delay", interpolate=True)
    s2 = plan.add_search(0, "delay", pre - 300, pre + 300, interpolate=True, lowchi2_fcutoff=19000.0)
    s3 = plan.add_search(0, "delay", lowchi2_fcutoff=50000.0)
    s4 = plan.add_search(0, "delay", 0, 1, interpolate=True)
    s5 = plan.add_search(0, "nodelay", lowchi2_fcutoff=62000.0)
    s6 = plan.add_search(0, "delay", pre - 500, pre + 500, outside=True)
    s7 = plan.add_search(0, "delay", 0, 300)
    x, _, _ = synth.make_traces(41, tmpl, psd, FS, ft.ampres, seed=77, max_delay=N // 16)
    x32 = x.astype(np.float32)
    x64 = x32.astype(np.float64)
    out = _run(plan, x32)
    P = lambda *a, **k: orc.process_events(filt, x64, *a, **k)
    check_search(out, plan.search_offset(0, s1), P("unconstrained", interpolate=True), "", ft.ampres, FS,
                 "interp", interpolated=True)
    check_search(out, plan.search_offset(0, s2),
                 P("constrained", lowchi2_fcutoff=19000.0, interpolate=True, window_min_index=pre - 300,
                   window_max_index=pre + 300), "", ft.ampres, FS, "interp-win", interpolated=True)
    check_search(out, plan.search_offset(0, s3), P("unconstrained", lowchi2_fcutoff=50000.0), "", ft.ampres,
                 FS, "50 kHz")
    check_search(out, plan.search_offset(0, s4),
                 P("constrained", interpolate=True, window_min_index=0, window_max_index=1), "", ft.ampres,
                 FS, "interp-edge", interpolated=True)
    check_search(out, plan.search_offset(0, s5), P("nodelay", lowchi2_fcutoff=62000.0), "", ft.ampres, FS,
                 "nodelay 62 kHz")
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


def test_three_slots_windows_bands_and_channel_algebra():
    """The example YAML's shape at its own trace length: three template tags on one plan (one
    launch, shared forward transform), time-domain windows, psd_amp bands, a summed channel --
    against the oracle, and every slot bit-identical to a single-slot plan of its own."""
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
    wins = [(100, (9 * N) // 25), (0, N - 1), (N // 2 - 500, N // 2 + 1263), (1250, 2500)]
    wid = [plan.add_tdwindow(a, b) for a, b in wins]
    bands = [(1, 20), (N // 60, N // 36), (N // 25, N // 20)]
    bid = [plan.add_band(a, b) for a, b in bands]
    ev, _, _ = synth.make_traces(2 * 21, tmpls[0], psd, FS, fts[0].ampres, seed=314)
    ev = ev.reshape(21, 2, N).astype(np.float32)
    plan.set_channels(2, [0, 1], [0.75, -1.25])
    out = plan.process(torch.as_tensor(ev, device="cuda:0")).cpu().numpy().astype(np.float64)
    assert plan.engine == "fused"
    x64 = combine_fp32(ev, [0, 1], [0.75, -1.25])         # the trace the device forms, bit for bit
    for s, (ft, filt) in enumerate(zip(fts, filts)):
        r_nd = orc.process_events(filt, x64, "nodelay", lowchi2_fcutoff=50000.0)
        r_un = orc.process_events(filt, x64, "unconstrained", lowchi2_fcutoff=50000.0)
        r_co = orc.process_events(filt, x64, "constrained", window_min_index=pre - 500,
                                  window_max_index=pre + 500, interpolate=(s == 1))
        for j, r in enumerate((r_nd, r_un, r_co)):
            check_search(out, plan.search_offset(s, ids[s][j]), r, "", ft.ampres, FS, f"{kinds[s]} search {j}",
                         interpolated=(j == 2 and s == 1))
    for i, (a, b) in enumerate(wins):
        t = plan.tdwindow_offset(wid[i])
        sc = np.abs(x64).max()
        assert np.allclose(out[:, t + 0], orc.baseline(x64, a, b), rtol=1e-4, atol=1e-6 * sc), (a, b)
        assert np.allclose(out[:, t + 1], orc.integral(x64, FS, a, b), rtol=1e-4, atol=1e-6 * sc * (b - a) / FS), (a, b)
        assert np.allclose(out[:, t + 2], x64[:, a:b].max(axis=1), rtol=2e-6, atol=1e-7 * sc)
        assert np.allclose(out[:, t + 3], x64[:, a:b].min(axis=1), rtol=2e-6, atol=1e-7 * sc)
    for i, (lo, hi) in enumerate(bands):
        V = np.fft.rfft(x64, axis=-1)[:, lo:hi] / N
        want = np.sqrt(2.0 * np.abs(V) ** 2 * N / FS).mean(axis=-1)
        assert np.allclose(out[:, plan.band_offset(bid[i])], want, rtol=2e-5), (lo, hi)
    # single-slot plans on the combined trace: bit-identical search records
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


def test_edge_cases_and_fallback():
    import torch
    from detprocess_amd import _lib
    plan, ft, filt, tmpl, psd = _mk()
    sid = plan.add_search(0, "delay")
    plan.add_search(0, "nodelay")
    wid = plan.add_tdwindow(0, N - 1)
    x = np.zeros((6, N), dtype=np.float32)
    x[1] = 3e-8
    x[2, 12345] = 1e-7
    x[3] = (2e-7 * np.roll(tmpl, -4000)).astype(np.float32)
    x[4] = x[3]
    x[5] = x[3]
    x[5, 777] = np.nan
    out = _run(plan, x)
    o = plan.search_offset(0, sid)
    assert out[0, o + 7] == 0 and out[0, o + 0] == 0 and out[0, o + 2] == 0     # all-zero: first rolled bin
    ref = orc.process_events(filt, x[:5].astype(np.float64), "unconstrained")
    assert np.array_equal(out[2:5, o + 7].astype(int), ref["index"][2:])
    assert abs(out[1, o + 0]) < 1e-3 * ft.ampres
    assert out[3, o + 7] == N // 2 - 4000
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
    # beyond the 1250 stashed bins: AUTO falls back for that call (LDS / ROCFFT engine), FUSED refuses
    plan_a, ft, filt, tmpl, psd = _mk(engine="auto")
    sa = plan_a.add_search(0, "delay", lowchi2_fcutoff=80000.0)
    xs, _, _ = synth.make_traces(9, tmpl, psd, FS, ft.ampres, seed=3)
    x32 = xs.astype(np.float32)
    outa = _run(plan_a, x32)
    ref = orc.process_events(filt, x32.astype(np.float64), "unconstrained", lowchi2_fcutoff=80000.0)
    check_search(outa, plan_a.search_offset(0, sa), ref, "", ft.ampres, FS, "auto/80kHz")
    plan_f, *_ = _mk(engine="fused")
    plan_f.add_search(0, "delay", lowchi2_fcutoff=80000.0)
    with pytest.raises(_lib.OfxError):
        _run(plan_f, x32)


def test_agrees_with_the_lds_engine_at_scale():
    """16384 generated traces: same t0 bins as the LDS engine, amplitudes and chi2 within the
    fp32 error of either engine."""
    import torch
    plan, ft, filt, tmpl, psd = _mk()
    pl2, *_ = _mk(engine="lds")
    plan.add_search(0, "delay"); pl2.add_search(0, "delay")
    from detprocess_amd import synth_traces
    B = 16384
    x, _ = synth_traces(B, N, tmpl, 0.0, 30 * ft.ampres, 300 * ft.ampres, 0.5, N // 16, seed=5, psd=psd, fs=FS)
    a = plan.process(x).cpu().numpy().astype(np.float64)
    b = pl2.process(x).cpu().numpy().astype(np.float64)
    flips = np.nonzero(a[:, 7] != b[:, 7])[0]
    assert len(flips) <= 2, f"{len(flips)} t0 bins differ"
    keep = np.ones(B, bool); keep[flips] = False
    assert np.all(np.abs(a[keep, 0] - b[keep, 0]) <= 2e-5 * np.abs(b[keep, 0]) + 2e-4 * ft.ampres)
    assert np.all(np.abs(a[keep, 2] - b[keep, 2]) <= 2e-5 * np.abs(b[keep, 2]) + 4e-6 * b[keep, 4])
    assert np.allclose(a[:, 4], b[:, 4], rtol=1e-5)


@pytest.mark.parametrize("feat", range(8))
@pytest.mark.parametrize("nslots", [1, 2, 3])
def test_every_kernel_instantiation_vs_oracle(feat, nslots):
    """One plan per instantiation of the kernel -- FEAT bit 0 a windowed fit, bit 1 time-domain
    windows, bit 2 channel algebra, one / several filter slots -- with every search of every slot
    and every window checked against the oracle on every event (the fuzz found an instantiation,
    windows + several slots without a windowed fit, that no other test built)."""
    import torch
    from detprocess_amd import OFPlan
    pre, B = N // 2 + 37, 48
    kinds = ("pulse", "glitch", "muon")[:nslots]
    psd = synth.make_psd(N, FS)
    tmpls = [synth.make_template(N, pre, FS, k) for k in kinds]
    fts = [build_filter(t, psd, FS, pre) for t in tmpls]
    filts = [orc.OFFilter(t, psd, FS, pre) for t in tmpls]
    plan = OFPlan(N, pre, FS, max_batch=64, device=0, engine="fused")
    ids = []
    for s, ft in enumerate(fts):
        plan.set_filter(s, ft)
        ss = [("nodelay", plan.add_search(s, "nodelay")), ("unconstrained", plan.add_search(s, "delay"))]
        if feat & 1:
            ss.append(("constrained", plan.add_search(s, "delay", pre - 400, pre + 400)))
        ids.append(ss)
    wins = [(N // 10, N // 2), (N // 2 - 300, N // 2 + 900)] if feat & 2 else []
    wid = [plan.add_tdwindow(a, b) for a, b in wins]
    nch = 2 if feat & 4 else 1
    ev, _, _ = synth.make_traces(B * nch, tmpls[0], psd, FS, fts[0].ampres, seed=5 + feat, max_delay=N // 16)
    ev = ev.reshape(B, nch, N).astype(np.float32)
    if feat & 4:
        plan.set_channels(2, [1, 0], [1.0, -0.5])
        x64 = combine_fp32(ev, [1, 0], [1.0, -0.5])       # the trace the device forms, bit for bit
        out = plan.process(torch.as_tensor(ev, device="cuda:0")).cpu().numpy().astype(np.float64)
    else:
        x64 = ev[:, 0].astype(np.float64)
        out = plan.process(torch.as_tensor(ev[:, 0], device="cuda:0")).cpu().numpy().astype(np.float64)
    for s, (ft, filt) in enumerate(zip(fts, filts)):
        for mode, sid in ids[s]:
            kw = dict(window_min_index=pre - 400, window_max_index=pre + 400) if mode == "constrained" else {}
            r = orc.process_events(filt, x64, mode, **kw)
            check_search(out, plan.search_offset(s, sid), r, "", ft.ampres, FS, f"k_fused25<{feat}> x{nslots} slot {s} {mode}")
    sc = np.abs(x64).max()
    for (a, b), w in zip(wins, wid):
        t_ = plan.tdwindow_offset(w)
        assert np.allclose(out[:, t_ + 0], orc.baseline(x64, a, b), rtol=1e-4, atol=1e-6 * sc)
        assert np.allclose(out[:, t_ + 2], x64[:, a:b].max(axis=1), rtol=2e-6, atol=1e-7 * sc)
        assert np.allclose(out[:, t_ + 3], x64[:, a:b].min(axis=1), rtol=2e-6, atol=1e-7 * sc)
