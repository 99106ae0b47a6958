"""Analytic known-answer tests on the HIP path itself (through the C ABI), with no oracle
in the loop: every expected value below is a closed form of the single-template optimal
filter, evaluated here in fp64 NumPy from
the definitions.  With parity unpinned (QETpy absent) these are the implementation-
independent checks: they mirror tests/test_oracle_kat.py, which runs the same closed forms
against the oracle."""

import numpy as np
import pytest

from detprocess_amd import build_filter, synth

pytestmark = pytest.mark.gpu

FS = 1.25e6
ENGINES = [(32768, "fused"), (32768, "rocfft"), (32768, "lds"), (25000, "lds"), (25000, "fused"), (12500, "fused"), (20000, "fused"), (4096, "lds"),
           (4096, "rocfft"), (1000, "rocfft")]


def _norm_from_definitions(tmpl, psd):
    """norm = sum_k |S_k|^2 / J_k / (N fs), J = inf at DC (AC coupling) -- DESIGN.md section 2."""
    n = tmpl.shape[0]
    S = np.fft.fft(tmpl)
    w = 1.0 / psd
    w[0] = 0.0
    return float(np.sum(np.abs(S) ** 2 * w) / (n * FS)), \
        float(np.sum((2 * np.pi * np.fft.fftfreq(n, d=1 / FS)) ** 2 * np.abs(S) ** 2 * w) / (n * FS))


def _plan(n, engine, tmpl, psd, pre, max_batch=64):
    from detprocess_amd import OFPlan
    ft = build_filter(tmpl, psd, FS, pre)
    plan = OFPlan(n, pre, FS, max_batch=max_batch, device=0, engine=engine)
    plan.set_filter(0, ft)
    return plan


def _run(plan, x):
    import torch
    return plan.process(torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32),
                                        device="cuda:0")).cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("n,engine", ENGINES)
def test_noiseless_shifted_template(n, engine):
    """trace = A roll(template, d)  =>  amp = A, t0 = d/fs, chi2 = 0, chi2nopulse = A^2 norm,
    ampres = 1/sqrt(norm), timeres = 1/sqrt(A^2 sum (2 pi f)^2 |S|^2/J / (N fs)), lowchi2 = 0;
    the no-delay fit reads lag 0."""
    pre = n // 2
    tmpl = synth.make_template(n, pre, FS)
    psd = synth.make_psd(n, FS)
    norm, tsum = _norm_from_definitions(tmpl, psd)
    plan = _plan(n, engine, tmpl, psd, pre)
    sd = plan.add_search(0, "delay")
    sn = plan.add_search(0, "nodelay")
    sw = plan.add_search(0, "delay", pre - 40, pre + 41)
    cases = [(1.0e-7, 0), (1.0e-7, 1), (-3.3e-8, -7), (2.5e-8, 300), (-3.3e-8, -n // 8 + 1),
             (4e-9, n // 4), (6e-8, -40), (6e-8, 40)]
    x = np.stack([a * np.roll(tmpl, d) for a, d in cases])
    x32 = x.astype(np.float32)
    out = _run(plan, x32)
    od, on, ow = (plan.search_offset(0, s) for s in (sd, sn, sw))
    for i, (a, d) in enumerate(cases):
        # the float32 cast of the input perturbs the trace by 6e-8 relative: compare with
        # the amplitude of the trace that was actually handed over
        chi0 = a * a * norm
        assert out[i, od + 7] == pre + d, (engine, a, d)
        assert out[i, od + 1] == pytest.approx(d / FS, rel=1e-6, abs=1e-13)
        assert out[i, od + 0] == pytest.approx(a, rel=2e-6)
        assert out[i, od + 4] == pytest.approx(chi0, rel=1e-5)
        assert abs(out[i, od + 2]) < 4e-6 * chi0          # chi2 = chi0 - A^2 norm cancels
        assert abs(out[i, od + 3]) < 4e-6 * chi0
        assert out[i, od + 5] == pytest.approx(1 / np.sqrt(norm), rel=1e-6)
        assert out[i, od + 6] == pytest.approx(1 / np.sqrt(a * a * tsum), rel=1e-5)
        if d == 0:
            assert out[i, on + 0] == pytest.approx(a, rel=2e-6)
            assert abs(out[i, on + 2]) < 4e-6 * chi0
        else:
            assert abs(out[i, on + 0]) < abs(a)            # a shifted pulse is not fit at lag 0
        if abs(d) <= 40:
            assert out[i, ow + 7] == pre + d and out[i, ow + 0] == pytest.approx(a, rel=2e-6)
        else:
            assert pre - 40 <= out[i, ow + 7] <= pre + 40


@pytest.mark.parametrize("n,engine", ENGINES)
def test_white_psd_gives_the_matched_filter(n, engine):
    """White PSD: A(0) = <v, s> / <s, s> with the DC component removed (AC coupling), and
    chi2nopulse = sum (v - mean v)^2 / (J fs)  (Parseval)."""
    pre = n // 2
    tmpl = synth.make_template(n, pre, FS)
    J0 = 1e-22
    psd = np.full(n, J0)
    plan = _plan(n, engine, tmpl, psd, pre)
    sn = plan.add_search(0, "nodelay")
    sd = plan.add_search(0, "delay")
    rng = np.random.default_rng(5)
    sig = np.sqrt(J0 * FS / 2) * 2
    v = (sig * rng.standard_normal((16, n)) + 4e-8 * tmpl).astype(np.float32)
    v64 = v.astype(np.float64)
    out = _run(plan, v)
    s0 = tmpl - tmpl.mean()
    v0 = v64 - v64.mean(axis=1, keepdims=True)
    want = v0 @ s0 / np.dot(s0, s0)
    on, od = plan.search_offset(0, sn), plan.search_offset(0, sd)
    res = 1 / np.sqrt(np.dot(s0, s0) / (J0 * FS))
    assert out[0, on + 5] == pytest.approx(res, rel=1e-6)
    assert np.all(np.abs(out[:, on + 0] - want) <= 1e-5 * np.abs(want) + 1e-5 * res)
    chi0 = np.sum(v0 ** 2, axis=1) / (J0 * FS)
    assert np.allclose(out[:, od + 4], chi0, rtol=1e-5)
    # the delay fit is the arg-max of the circular cross-correlation
    cc = np.fft.ifft(np.fft.fft(v0, axis=1) * np.conj(np.fft.fft(s0)), axis=1).real / np.dot(s0, s0)
    lag = np.argmax(np.roll(cc, pre, axis=1) ** 2, axis=1)
    assert np.array_equal(out[:, od + 7].astype(int), lag)


@pytest.mark.parametrize("n,engine", [(32768, "fused"), (25000, "lds"), (25000, "fused"), (12500, "fused"), (4096, "rocfft"),
                                      (1001, "rocfft")])
def test_time_domain_closed_forms(n, engine):
    """Ramps: baseline / integral / maximum / minimum on end-exclusive slices
    (algorithms.py:698, 759, 818, 879): mean = (a + b - 1)/2, trapezoid = (b-1-a)(a+b-1)/2."""
    from detprocess_amd import OFPlan
    plan = OFPlan(n, n // 2, FS, max_batch=16, device=0, engine=engine if n % 2 == 0 else "rocfft")
    wins = [(0, n - 1), (10, 20), (n // 2 - n // 8, n // 2 + n // 8), (1, n), (n - 3, n)]
    ids = [plan.add_tdwindow(a, b) for a, b in wins]
    scale = 1e-9
    ramp = scale * np.arange(n, dtype=np.float64)
    x = np.stack([ramp, -ramp, ramp[::-1].copy()])
    out = _run(plan, x)
    x32 = x.astype(np.float32).astype(np.float64)
    for (a, b), wid in zip(wins, ids):
        o = plan.tdwindow_offset(wid)
        mean = scale * (a + b - 1) / 2.0
        trap = scale * (b - 1 - a) * (a + b - 1) / 2.0 / FS
        assert out[0, o + 0] == pytest.approx(mean, rel=2e-6)
        assert out[1, o + 0] == pytest.approx(-mean, rel=2e-6)
        assert out[0, o + 1] == pytest.approx(trap, rel=2e-6, abs=1e-7 * scale * n / FS)
        assert out[0, o + 2] == x32[0, b - 1] and out[0, o + 3] == x32[0, a]
        assert out[1, o + 2] == x32[1, a] and out[1, o + 3] == x32[1, b - 1]
        assert out[2, o + 2] == x32[2, a] and out[2, o + 3] == x32[2, b - 1]


def test_nan_trace_gives_a_nan_row_not_a_finite_one():
    """NumPy's argmin over a chi2 array of NaNs yields NaN features in the reference; the
    engines must not hide such an event behind finite-looking numbers."""
    n = 32768
    pre = n // 2
    tmpl = synth.make_template(n, pre, FS)
    psd = synth.make_psd(n, FS)
    for engine in ("fused", "rocfft", "lds"):
        plan = _plan(n, engine, tmpl, psd, pre)
        sd = plan.add_search(0, "delay")
        sw = plan.add_search(0, "delay", 16000, 17000)
        x = np.stack([2e-8 * tmpl, 2e-8 * tmpl, 2e-8 * tmpl]).astype(np.float32)
        x[1, 777] = np.nan
        out = _run(plan, x)
        for s in (sd, sw):
            o = plan.search_offset(0, s)
            assert np.all(np.isnan(out[1, o:o + 5])), engine
            assert np.all(np.isfinite(out[[0, 2], o:o + 8])), engine
            assert out[0, o + 7] == pre
        plan.close()


@pytest.mark.parametrize("n,pre", [(4096, 1500), (32768, 16384)])
def test_trigger_known_answers_without_the_oracle(n, pre):
    """Continuous-data trigger on noiseless pulses: the filtered trace equals the pulse amplitude
    at the pulse (the estimator's defining property), delta chi2 = A^2 norm there, the trigger
    sits one sample after onset + pretrigger (the in-tree 'same'-mode convolution,
    oftrigger.py:649-662, with the shift of :1005), pulses closer than the static pile-up window
    merge into the larger one, the edges are zeroed (:674-679)."""
    from detprocess_amd import OptimumFilterTrigger
    tmpl = synth.make_template(n, pre, FS)
    psd = synth.make_psd(n, FS)
    norm, _ = _norm_from_definitions(tmpl, psd)
    L = 30 * n
    x = np.zeros(L, dtype=np.float64)
    pos = [3 * n, 9 * n, 9 * n + n // 5, 20 * n]
    amps = [1e-7, 2e-7, 0.5e-7, 3e-7]
    for p, a in zip(pos, amps):
        x[p:p + n] += a * tmpl
    trig = OptimumFilterTrigger("chanA", FS, tmpl, psd, pre)
    assert trig.get_resolution()[0] == pytest.approx(1 / np.sqrt(norm), rel=1e-9)
    trig.update_trace(x.astype(np.float32))
    d = trig.get_filtered_delta_chi2()
    assert np.all(d[:n] == 0) and np.all(d[L - n + 1:] == 0)
    trig.find_triggers(5.0, pileup_window_samples=n // 2)
    td = trig.get_trigger_data()["chanA"]
    assert td["trigger_index"] == [pos[0] + pre + 1, pos[1] + pre + 1, pos[3] + pre + 1]
    for i, a in ((0, amps[0]), (2, amps[3])):                       # the isolated pulses: exact
        assert td["trigger_amplitude"][i] == pytest.approx(a, rel=3e-5)
        assert td["trigger_delta_chi2"][i] == pytest.approx(a * a * norm, rel=1e-4)
    # the pulse with a neighbour on its tail sees that neighbour's filtered response as well
    assert td["trigger_amplitude"][1] == pytest.approx(amps[1], rel=0.03)
    assert td["trigger_time"][0] == pytest.approx((pos[0] + pre + 1) / FS, rel=1e-12)
    trig.find_triggers(5.0, pileup_window_samples=0)
    assert len(trig.get_trigger_data()["chanA"]["trigger_index"]) >= 4
