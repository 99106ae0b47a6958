"""Host-side mirror of the reference interface: YAML surface, channel
expressions, window indices, filter-file model (CPU), and on the GPU the
FeatureExtractors static methods and the batched FeatureProcessing driver."""
import numpy as np
import pytest

from detprocess_amd import FilterData, YamlConfig, synth, utils
from detprocess_amd.ofbase import search_range

FS = 1.25e6
CHANS = ["Melange1pc1ch", "Melange025pcLeft", "Melange025pcRight", "Melange4pc1ch"]

YAML = """
filter_file: /path/to/filter_file.hdf5
global:
    trace_length_msec: 26.2144
    pretrigger_length_msec: 13.1072
Melange1pc1ch:
    of1x1_nodelay:
        run: True
        lowchi2_fcutoff: 15000
        template_tag: default
        noise_tag: default
    of1x1_unconstrained:
        run: True
        template_tag: default
    of1x1_constrained:
        run: True
        window_min_from_trig_usec: -400
        window_max_from_trig_usec: 400
        template_tag: default
        csd_tag: default
    of1x1_glitch:
        run: True
        base_algorithm: of1x1_unconstrained
        template_tag: glitch
    of1x1_interp:
        run: True
        base_algorithm: of1x1_unconstrained
        template_tag: default
        interpolate: True
    baseline:
        run: True
        window_min_from_start_usec: 0
        window_max_from_trig_usec: -2000
    baseline_end:
        run: True
        base_algorithm: baseline
        window_min_from_trig_usec: 2000
        window_max_to_end_usec: 0
    maximum:
        run: True
        window_min_from_trig_usec: -500
        window_max_from_trig_usec: 500
    minimum:
        run: True
        window_min_from_trig_usec: -500
        window_max_from_trig_usec: 500
    integral:
        run: True
        window_min_from_trig_usec: -10
        window_max_from_trig_usec: 500
    energyabsorbed:
        run: False
        i0: 88e-9
Melange025pcLeft+Melange025pcRight:
    feature_channel: MelangeSum
    weight_Melange025pcLeft: 0.9
    weight_Melange025pcRight: 1.1
    of1x1_unconstrained:
        run: True
        template_tag: default
    integral:
        run: True
Melange4pc1ch:
    disable: True
    baseline:
        run: True
"""


def test_yaml_surface():
    cfg = YamlConfig(YAML, CHANS, sample_rate=FS).get_config("feature")
    assert cfg["overall"]["filter_file"] == "/path/to/filter_file.hdf5"
    assert set(cfg["channels"]) == {"Melange1pc1ch", "Melange025pcLeft+Melange025pcRight"}
    a = cfg["channels"]["Melange1pc1ch"]
    assert "energyabsorbed" not in a                          # run: False dropped
    assert a["of1x1_nodelay"]["csd_tag"] == "default"          # noise_tag renamed
    assert a["of1x1_nodelay"]["nb_samples"] == 32768
    assert a["of1x1_nodelay"]["nb_pretrigger_samples"] == 16384
    assert cfg["traces_config"] == {(32768, 16384): ["Melange1pc1ch", "Melange025pcLeft",
                                                     "Melange025pcRight"]}
    assert cfg["weights"] == {"Melange025pcLeft+Melange025pcRight":
                              {"weight_Melange025pcLeft": 0.9, "weight_Melange025pcRight": 1.1}}
    assert cfg["channel_list"] == ["Melange1pc1ch", "Melange025pcLeft", "Melange025pcRight"]
    # comma lists and 'all' expand; duplicates are an error; run is required
    c2 = YamlConfig("A,B:\n  baseline:\n    run: True\n", ["A", "B"]).get_config("feature")
    assert set(c2["channels"]) == {"A", "B"}
    c3 = YamlConfig("all:\n  baseline:\n    run: True\n", ["A", "B", "C"]).get_config("feature")
    assert set(c3["channels"]) == {"A", "B", "C"}
    with pytest.raises(ValueError):
        YamlConfig("A:\n  baseline:\n    run: True\n  baseline:\n    run: True\n", ["A"])
    with pytest.raises(ValueError):
        YamlConfig("A:\n  baseline:\n    window_min_index: 3\n", ["A"])
    with pytest.raises(ValueError):
        YamlConfig("global:\n  trace_length_samples: 100\nA:\n  baseline:\n    run: True\n", ["A"])
    with pytest.raises(ValueError):
        YamlConfig("A+Q:\n  baseline:\n    run: True\n", ["A", "B"]).get_config("feature")


def test_split_channel_name():
    s = utils.split_channel_name
    assert s("A") == (["A"], None)
    assert s("A+B", ["A", "B"]) == (["A", "B"], "+")
    assert s("A|B", ["A", "B"]) == (["A", "B"], "|")
    assert s("A,B", separator=",") == (["A", "B"], ",")
    assert s("A-B", ["A", "B"]) == (["A", "B"], "-")
    assert s("Det-1", ["Det-1", "Det-2"]) == (["Det-1"], None)
    assert s("Det-1-Det-2", ["Det-1", "Det-2"])[0] == ["Det-1", "Det-2"]
    with pytest.raises(ValueError):
        s("A-B", separator="-")
    with pytest.raises(ValueError):
        s("A+Q", ["A", "B"])
    with pytest.raises(ValueError):
        s("A;B", ["A", "B"], separator=";")


def test_search_range_policies():
    assert search_range(32768, 16384, FS, -400, 400, 15884, 16884) == (15884, 16884)
    assert search_range(32768, 16384, FS, None, None, 15884, 16884) == (15884, 16884)
    assert search_range(32768, 16384, FS, None, None, 15884, 16884, "index") == (15884, 16885)
    assert search_range(32768, 16384, FS, -10, 10) == (16371, 16397)     # floor / ceil of +-12.5
    assert search_range(32768, 16384, FS) == (0, 32768)
    assert search_range(32768, 16384, FS, -1e9, 1e9) == (0, 32768)


def test_filterdata_model(tmp_path):
    n = 4096
    fd = FilterData()
    t = synth.make_template(n, n // 2, FS)
    J = synth.make_psd(n, FS)
    f = np.fft.fftfreq(n, d=1 / FS)
    fd.set_template("A", t, sample_rate=FS, pretrigger_length_samples=n // 2, tag="default")
    fd.set_psd("A", J, f, sample_rate=FS, tag="default")
    tt, time, meta = fd.get_template("A", return_metadata=True)
    assert np.array_equal(tt, t) and meta["nb_pretrigger_samples"] == n // 2
    assert time[n // 2] == 0 and time[0] == pytest.approx(-(n // 2) / FS)
    csd, freqs, m = fd.get_csd("A", fold=False, return_metadata=True)
    assert np.array_equal(csd, J) and m["sample_rate"] == FS
    pf, ff = fd.get_psd("A", fold=True)[:2]
    assert pf.shape[0] == n // 2 + 1 and pf[1] == pytest.approx(2 * J[1]) and pf[0] == J[0]
    with pytest.raises(ValueError):
        fd.set_psd("A", J[: n // 2 + 1], np.fft.rfftfreq(n, d=1 / FS))     # folded: refused
    with pytest.raises(ValueError):
        fd.get_template("B")
    with pytest.raises(ValueError):
        fd.get_template("A", tag="nope")
    fd.save_npz(tmp_path / "filt.npz")
    fd2 = FilterData()
    fd2.load_npz(tmp_path / "filt.npz")
    t2, _, m2 = fd2.get_template("A", return_metadata=True)
    assert np.array_equal(t2, t) and m2["nb_pretrigger_samples"] == n // 2


# ------------------------------------------------------------------- GPU side
def _filter_data(n, pre):
    fd = FilterData()
    f = np.fft.fftfreq(n, d=1 / FS)
    J = synth.make_psd(n, FS)
    for ch in CHANS + ["Melange025pcLeft+Melange025pcRight"]:
        fd.set_template(ch, synth.make_template(n, pre, FS, "pulse"), sample_rate=FS,
                        pretrigger_length_samples=pre, tag="default")
        fd.set_template(ch, synth.make_template(n, pre, FS, "glitch"), sample_rate=FS,
                        pretrigger_length_samples=pre, tag="glitch")
        fd.set_psd(ch, J, f, sample_rate=FS, tag="default")
    return fd, J


@pytest.mark.gpu
def test_feature_extractors_static_methods():
    from detprocess_amd import FeatureExtractors as FE, OFBase
    from oracle import of1x1 as orc
    n, pre = 32768, 16384
    tmpl = synth.make_template(n, pre, FS)
    J = synth.make_psd(n, FS)
    filt = orc.OFFilter(tmpl, J, FS, pre)
    x, _, _ = synth.make_traces(5, tmpl, J, FS, filt.ampres, seed=21)
    x32 = x.astype(np.float32)
    ob = OFBase(FS)
    ob.set_csd("chanA", J, coupling="AC")
    ob.add_template("chanA", tmpl, template_tag="default", pretrigger_samples=pre)
    assert ob.phi("chanA", "default") is None
    ob.calc_phi("chanA", "default")
    assert ob.phi("chanA", "default") is not None
    # no signal -> sentinel (algorithms.py:398-407)
    r = FE.of1x1_unconstrained("chanA", ob, template_tag="default")
    assert r == {"amp_of1x1_unconstrained": -999999.0, "t0_of1x1_unconstrained": -999999.0,
                 "chi2_of1x1_unconstrained": -999999.0, "lowchi2_of1x1_unconstrained": -999999.0}
    with pytest.raises(ValueError):
        FE.of1x1_nodelay("chanA", ob)                       # template tag required
    # single event -> scalars
    ob.update_signal("chanA", x32[0], calc_fft=True)
    ob.calc_signal_filt("chanA")
    ob.calc_signal_filt_td("chanA")
    r = FE.of1x1_constrained("chanA", ob, template_tag="default",
                             window_min_from_trig_usec=-400, window_max_from_trig_usec=400,
                             window_min_index=15884, window_max_index=16884,
                             feature_base_name="myof", fs=FS, nb_samples=n, junk=1)
    ref = orc.of1x1_withdelay(filt, x32[0].astype(np.float64), -400, 400)
    assert set(r) == {f"{q}_myof" for q in ("amp", "t0", "chi2", "lowchi2", "chi2nopulse",
                                            "ampres", "timeres")}
    assert isinstance(r["amp_myof"], float)
    assert r["amp_myof"] == pytest.approx(ref["amp"], rel=1e-5, abs=1e-4 * filt.ampres)
    assert r["t0_myof"] == pytest.approx(ref["t0"], rel=1e-6, abs=1e-12)
    assert r["chi2_myof"] == pytest.approx(ref["chi2"], rel=1e-5, abs=2e-6 * ref["chi2nopulse"])
    assert r["ampres_myof"] == pytest.approx(filt.ampres, rel=1e-6)
    # batch -> arrays
    ob.clear_signal()
    assert not ob.is_signal_stored("chanA")
    ob.update_signal("chanA", x32)
    r = FE.of1x1_nodelay("chanA", ob, template_tag="default", lowchi2_fcutoff=15000)
    refn = orc.process_events(filt, x32.astype(np.float64), "nodelay", lowchi2_fcutoff=15000)
    assert np.allclose(r["amp_of1x1_nodelay"], refn["amp"], rtol=1e-5, atol=1e-4 * filt.ampres)
    assert np.allclose(r["lowchi2_of1x1_nodelay"], refn["lowchi2"], rtol=1e-5, atol=1e-2)
    assert "t0_of1x1_nodelay" not in r                      # algorithms.py:344-348
    # interpolate=True (algorithms.py:357): refined t0 within half a bin of the discrete one
    ri = FE.of1x1_unconstrained("chanA", ob, template_tag="default", interpolate=True)
    refi = orc.process_events(filt, x32.astype(np.float64), "unconstrained", interpolate=True)
    assert np.allclose(ri["t0_of1x1_unconstrained"], refi["t0"], rtol=0, atol=1e-3 / FS)
    assert np.allclose(ri["amp_of1x1_unconstrained"], refi["amp"], rtol=1e-5,
                       atol=1e-4 * filt.ampres)
    # trace family
    tr = x32[1]
    assert FE.baseline(tr, 100, 5000)["baseline"] == pytest.approx(
        float(orc.baseline(tr.astype(np.float64), 100, 5000)), rel=1e-4, abs=1e-12)
    assert FE.integral(tr, FS, 16000, 17000, feature_base_name="int2")["int2"] == pytest.approx(
        float(orc.integral(tr.astype(np.float64), FS, 16000, 17000)), rel=1e-4, abs=1e-15)
    assert FE.maximum(tr)["maximum"] == float(tr[:-1].max())       # default max = N-1, exclusive
    assert FE.minimum(x32)["minimum"].shape == (5,)
    assert FE.baseline(np.array([]))["baseline"] == -999999.0
    odd = np.arange(1001, dtype=np.float32)
    assert FE.maximum(odd)["maximum"] == 999.0


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["auto", "rocfft"])
def test_feature_processing_batch_driver(engine):
    from detprocess_amd import FeatureProcessing
    from oracle import of1x1 as orc
    n, pre, B = 32768, 16384, 9
    fd, J = _filter_data(n, pre)
    tmpl = synth.make_template(n, pre, FS)
    filt = orc.OFFilter(tmpl, J, FS, pre)
    filt_g = orc.OFFilter(synth.make_template(n, pre, FS, "glitch"), J, FS, pre)
    ev, _, _ = synth.make_traces(B * 4, tmpl, J, FS, filt.ampres, seed=31)
    ev = ev.reshape(B, 4, n).astype(np.float32)
    fp = FeatureProcessing(YAML, fd, CHANS, FS, engine=engine)
    valid = np.ones(B, dtype=np.uint8)
    valid[4] = 0
    df = fp.process(ev, valid=valid)
    want_cols = {"amp_of1x1_nodelay_Melange1pc1ch", "chi2_of1x1_nodelay_Melange1pc1ch",
                 "lowchi2_of1x1_nodelay_Melange1pc1ch", "t0_of1x1_unconstrained_Melange1pc1ch",
                 "timeres_of1x1_constrained_Melange1pc1ch", "amp_of1x1_glitch_Melange1pc1ch",
                 "baseline_Melange1pc1ch", "baseline_end_Melange1pc1ch",
                 "maximum_Melange1pc1ch", "minimum_Melange1pc1ch", "integral_Melange1pc1ch",
                 "amp_of1x1_unconstrained_MelangeSum", "integral_MelangeSum"}
    assert want_cols <= set(df.columns)
    assert "t0_of1x1_nodelay_Melange1pc1ch" not in df.columns
    assert (df.iloc[4] == -999999.0).all()
    ok = valid.astype(bool)
    x = ev[:, 0, :].astype(np.float64)
    r = orc.process_events(filt, x, "unconstrained")
    assert np.array_equal(np.round(df["t0_of1x1_unconstrained_Melange1pc1ch"][ok] * FS),
                          (r["index"] - pre)[ok])
    assert np.allclose(df["amp_of1x1_unconstrained_Melange1pc1ch"][ok], r["amp"][ok], rtol=1e-5,
                       atol=1e-4 * filt.ampres)
    rc = orc.process_events(filt, x, "constrained", window_min_from_trig_usec=-400,
                            window_max_from_trig_usec=400)
    assert np.allclose(df["amp_of1x1_constrained_Melange1pc1ch"][ok], rc["amp"][ok], rtol=1e-5,
                       atol=1e-4 * filt.ampres)
    rn = orc.process_events(filt, x, "nodelay", lowchi2_fcutoff=15000)
    assert np.allclose(df["lowchi2_of1x1_nodelay_Melange1pc1ch"][ok], rn["lowchi2"][ok],
                       rtol=1e-5, atol=2e-6 * r["chi2nopulse"][ok].max())
    ri = orc.process_events(filt, x, "unconstrained", interpolate=True)
    assert np.allclose(df["t0_of1x1_interp_Melange1pc1ch"][ok], ri["t0"][ok], rtol=0, atol=1e-3 / FS)
    assert np.allclose(df["chi2_of1x1_interp_Melange1pc1ch"][ok], ri["chi2"][ok], rtol=1e-5,
                       atol=2e-6 * r["chi2nopulse"][ok].max())
    rg = orc.process_events(filt_g, x, "unconstrained")
    assert np.allclose(df["amp_of1x1_glitch_Melange1pc1ch"][ok], rg["amp"][ok], rtol=1e-5,
                       atol=1e-4 * filt_g.ampres)
    # trace features with the reference's window arithmetic and end-exclusive slices
    lo, hi = orc.get_window_indices(n, pre, FS, window_min_from_start_usec=0,
                                    window_max_from_trig_usec=-2000)
    assert np.allclose(df["baseline_Melange1pc1ch"][ok], orc.baseline(x, lo, hi)[ok], rtol=1e-4,
                       atol=1e-6 * np.abs(x).max())
    lo, hi = orc.get_window_indices(n, pre, FS, window_min_from_trig_usec=2000,
                                    window_max_to_end_usec=0)
    assert np.allclose(df["baseline_end_Melange1pc1ch"][ok], orc.baseline(x, lo, hi)[ok],
                       rtol=1e-4, atol=1e-6 * np.abs(x).max())
    lo, hi = orc.get_window_indices(n, pre, FS, window_min_from_trig_usec=-500,
                                    window_max_from_trig_usec=500)
    assert np.array_equal(df["maximum_Melange1pc1ch"][ok],
                          orc.maximum(x, lo, hi)[ok].astype(np.float32).astype(np.float64))
    lo, hi = orc.get_window_indices(n, pre, FS, window_min_from_trig_usec=-10,
                                    window_max_from_trig_usec=500)
    assert np.allclose(df["integral_Melange1pc1ch"][ok], orc.integral(x, FS, lo, hi)[ok],
                       rtol=1e-4, atol=1e-6 * np.abs(x).max() * (hi - lo) / FS)
    # summed channel with weights
    xs = 0.9 * ev[:, 1, :].astype(np.float64) + 1.1 * ev[:, 2, :].astype(np.float64)
    rs = orc.process_events(filt, xs, "unconstrained")
    assert np.allclose(df["amp_of1x1_unconstrained_MelangeSum"][ok], rs["amp"][ok], rtol=3e-5,
                       atol=2e-4 * filt.ampres)
    assert np.allclose(df["integral_MelangeSum"][ok], orc.integral(xs, FS, 0, n - 1)[ok],
                       rtol=1e-4, atol=1e-6 * np.abs(xs).max() * n / FS)


EXT_FILE = '''
import numpy as np
import detprocess_amd as da


class FeatureExtractors:
    @staticmethod
    def minmax(trace, window_min_index=None, window_max_index=None, feature_base_name="minmax",
               **kwargs):
        if window_min_index is None:
            window_min_index = 0
        if window_max_index is None:
            window_max_index = trace.shape[-1] - 1
        w = trace[window_min_index:window_max_index]
        return {feature_base_name: np.amax(w) - np.amin(w)}

    @staticmethod
    def of1x1_twice(channel, of_base, template_tag="default", feature_base_name="of1x1_twice",
                    **kwargs):
        r = da.FeatureExtractors.of1x1_unconstrained(channel, of_base, template_tag=template_tag,
                                                     feature_base_name="tmp")
        return {"amp2_" + feature_base_name: 2.0 * np.asarray(r["amp_tmp"])}
'''


@pytest.mark.gpu
def test_external_extractor_file(tmp_path):
    """Plugin mechanism (features.py:248-263, 1002-1029, 1105-1131): user algorithms from
    an external file, trace-based and OF-based, with the injected kwargs and column naming
    of the reference; duplicates of built-ins are rejected."""
    from detprocess_amd import FeatureProcessing
    from oracle import of1x1 as orc
    n, pre, B = 32768, 16384, 5
    fd, J = _filter_data(n, pre)
    tmpl = synth.make_template(n, pre, FS)
    filt = orc.OFFilter(tmpl, J, FS, pre)
    ev, _, _ = synth.make_traces(B * 4, tmpl, J, FS, filt.ampres, seed=12)
    ev = ev.reshape(B, 4, n).astype(np.float32)
    ext = tmp_path / "features_user.py"
    ext.write_text(EXT_FILE)
    yaml_text = YAML.replace("    energyabsorbed:\n        run: False\n        i0: 88e-9\n",
                             "    minmax:\n        run: True\n        window_min_from_trig_usec: -500\n"
                             "        window_max_from_trig_usec: 500\n"
                             "    mm_all:\n        run: True\n        base_algorithm: minmax\n"
                             "    of1x1_twice:\n        run: True\n        template_tag: default\n")
    assert "minmax" in yaml_text
    fp = FeatureProcessing(yaml_text, fd, CHANS, FS, external_file=str(ext))
    valid = np.array([1, 1, 0, 1, 1], dtype=np.uint8)
    df = fp.process(ev, valid=valid)
    ok = valid.astype(bool)
    x = ev[:, 0, :].astype(np.float64)
    lo, hi = orc.get_window_indices(n, pre, FS, window_min_from_trig_usec=-500,
                                    window_max_from_trig_usec=500)
    want = np.array([x[b, lo:hi].max() - x[b, lo:hi].min() for b in range(B)])
    assert np.allclose(df["minmax_Melange1pc1ch"][ok], want[ok], rtol=1e-12)
    want_all = np.array([x[b, 0:n - 1].max() - x[b, 0:n - 1].min() for b in range(B)])
    assert np.allclose(df["mm_all_Melange1pc1ch"][ok], want_all[ok], rtol=1e-12)
    assert df["minmax_Melange1pc1ch"][2] == -999999.0
    assert np.allclose(df["amp2_of1x1_twice_Melange1pc1ch"][ok],
                       2.0 * df["amp_of1x1_unconstrained_Melange1pc1ch"][ok], rtol=1e-6)
    assert df["amp2_of1x1_twice_Melange1pc1ch"][2] == -999999.0
    dup = tmp_path / "dup.py"
    dup.write_text("class FeatureExtractors:\n    @staticmethod\n    def baseline(trace, **kw):\n"
                   "        return {}\n")
    with pytest.raises(ValueError, match="duplicate"):
        FeatureProcessing(yaml_text, fd, CHANS, FS, external_file=str(dup))
    with pytest.raises(ValueError, match="Cannot find algorithm"):
        FeatureProcessing(yaml_text, fd, CHANS, FS).process(ev)


YAML_25000 = """
filter_file: /path/to/filter_file.hdf5
global:
    trace_length_msec: 20
    pretrigger_length_msec: 10
Melange1pc1ch:
    of1x1_nodelay:
        run: True
        template_tag: default
    of1x1_constrained:
        run: True
        template_tag: default
        window_min_from_trig_usec: -400
        window_max_from_trig_usec: 400
    baseline:
        run: True
        window_min_from_start_usec: 0
        window_max_from_trig_usec: -2000
    integral:
        run: True
        window_min_from_trig_usec: -10
        window_max_from_trig_usec: 500
    psd_amp:
        run: True
        f_lims: [[50, 100], [2000, 4000]]
Melange025pcLeft,Melange025pcRight:
    of1x1_unconstrained:
        run: True
        template_tag: default
        interpolate: True
    maximum:
        run: True
Melange025pcLeft+Melange025pcRight:
    feature_channel: MelangeSum
    weight_Melange025pcLeft: 0.9
    weight_Melange025pcRight: 1.1
    of1x1_constrained:
        run: True
        template_tag: default
        window_min_from_trig_usec: -400
        window_max_from_trig_usec: 400
Melange025pcLeft|Melange025pcRight:
    feature_channel: MelangeLR
    of2x2:
        run: True
        base_algorithm: ofnxm
        amplitude_names: [ampshared, ampslow]
        window_min_from_trig_usec: -100
        window_max_from_trig_usec: 100
        csd_tag: default
        template_tag: pair
    of1x2x2_test:
        run: True
        base_algorithm: of1x2x2
        template_tag: pair
"""


@pytest.mark.gpu
def test_example_shaped_config_at_25000_samples():
    """A configuration shaped like the reference's example (20 ms traces at 1.25 MHz = 25000
    samples, comma-separated channel blocks, a summed channel, psd_amp, a 2x2 ``a|b`` block
    and an algorithm outside this engine): of1x1 runs on the FUSED engine (k_fused25), ofnxm on
    the NxM engine, through the YAML driver."""
    from detprocess_amd import FeatureProcessing
    from oracle import of1x1 as orc
    from oracle import ofnxm as onm
    from test_ofnxm import make_csd, make_templates
    n, pre, B = 25000, 12500, 7
    fd, J = _filter_data(n, pre)
    pair = "Melange025pcLeft|Melange025pcRight"
    t2 = make_templates(n, pre, 2, 2)
    c2 = make_csd(n, 2)
    fd.set_template(pair, t2, sample_rate=FS, pretrigger_length_samples=pre, tag="pair")
    fd.set_csd(pair, c2, np.fft.fftfreq(n, d=1 / FS), sample_rate=FS, tag="default")
    tmpl = synth.make_template(n, pre, FS)
    filt = orc.OFFilter(tmpl, J, FS, pre)
    ev, _, _ = synth.make_traces(B * 4, tmpl, J, FS, filt.ampres, seed=5, max_delay=n // 16)
    ev = ev.reshape(B, 4, n).astype(np.float32)
    with pytest.raises(NotImplementedError, match="of1x2x2"):
        FeatureProcessing(YAML_25000, fd, CHANS, FS).process(ev)
    with pytest.warns(UserWarning, match="outside the hot path"):
        fp = FeatureProcessing(YAML_25000, fd, CHANS, FS, skip_unsupported=True)
        df = fp.process(ev)
    assert {cp.plan.engine for cp in fp._plans.values() if not getattr(cp, "nxm", False)} == {"fused"}
    # the 2x2 block: windowed delay fit + no-delay fit, named as algorithms.py:229-273
    fn = onm.NxMFilter(t2, c2, FS, pre)
    rn = onm.process_events(fn, ev[:, 1:3, :].astype(np.float64), window_min_from_trig_usec=-100,
                            window_max_from_trig_usec=100)
    assert np.array_equal(np.round(df["t0_of2x2_constrained_MelangeLR"] * FS), rn["index"] - pre)
    for i, an in enumerate(("ampshared", "ampslow")):
        for kind, key in (("constrained", "amps"), ("nodelay", "amps_nodelay")):
            assert np.allclose(df[f"{an}_of2x2_{kind}_MelangeLR"], rn[key][:, i], rtol=2e-5,
                               atol=1e-4 * fn.ampres[i])
    for kind, key in (("constrained", "chi2"), ("nodelay", "chi2_nodelay")):
        assert np.allclose(df[f"chi2_of2x2_{kind}_MelangeLR"], rn[key], rtol=2e-5,
                           atol=2e-6 * rn["chi2_0"].max())
    assert not any("of1x2x2" in c for c in df.columns)
    x0 = ev[:, 0, :].astype(np.float64)
    rn = orc.process_events(filt, x0, "nodelay")
    assert np.allclose(df["amp_of1x1_nodelay_Melange1pc1ch"], rn["amp"], rtol=1e-5, atol=1e-4 * filt.ampres)
    rc = orc.process_events(filt, x0, "constrained", window_min_from_trig_usec=-400,
                            window_max_from_trig_usec=400)
    assert np.array_equal(np.round(df["t0_of1x1_constrained_Melange1pc1ch"] * FS), rc["index"] - pre)
    assert np.allclose(df["chi2_of1x1_constrained_Melange1pc1ch"], rc["chi2"], rtol=1e-5,
                       atol=2e-6 * rc["chi2nopulse"].max())
    for j, ch in ((1, "Melange025pcLeft"), (2, "Melange025pcRight")):
        ri = orc.process_events(filt, ev[:, j, :].astype(np.float64), "unconstrained", interpolate=True)
        assert np.allclose(df[f"t0_of1x1_unconstrained_{ch}"], ri["t0"], rtol=0, atol=1e-3 / FS)
        assert np.array_equal(df[f"maximum_{ch}"], ev[:, j, :-1].max(axis=1).astype(np.float64))
    xs = 0.9 * ev[:, 1, :].astype(np.float64) + 1.1 * ev[:, 2, :].astype(np.float64)
    rs = orc.process_events(filt, xs, "constrained", window_min_from_trig_usec=-400,
                            window_max_from_trig_usec=400)
    assert np.allclose(df["amp_of1x1_constrained_MelangeSum"], rs["amp"], rtol=3e-5, atol=2e-4 * filt.ampres)
    pa = orc.psd_amp(x0, FS, [[50, 100], [2000, 4000]])
    for name, v in pa.items():
        assert np.allclose(df[f"psd_amp_{name}_Melange1pc1ch"], v, rtol=1e-4)


@pytest.mark.gpu
def test_feature_processing_from_adc_streams():
    """The YAML-driven batch driver on events cut from continuous int16 streams equals the
    same driver on host-cut, host-converted events (4 channels, per-channel conversion)."""
    from detprocess_amd import FeatureProcessing
    n, pre = 32768, 16384
    fd, J = _filter_data(n, pre)
    tmpl = synth.make_template(n, pre, FS)
    ampres = 1.0 / np.sqrt(1.0)  # only a scale for the synthetic pulses
    from oracle import of1x1 as orc
    filt = orc.OFFilter(tmpl, J, FS, pre)
    x, _, _ = synth.make_traces(4 * 4, tmpl, J, FS, filt.ampres, seed=77)
    n_stream = 4 * n
    scale = np.array([2.0e-12, 2.5e-12, 3.0e-12, 1.5e-12])
    offset = np.array([-1e-9, 0.0, 2e-9, 5e-10])
    adc = np.empty((4, n_stream), dtype=np.int16)
    for c in range(4):
        train = x[4 * c:4 * c + 4].reshape(-1)
        adc[c] = np.clip(np.round((train - offset[c]) / scale[c]), -32768, 32767)
    trig = np.array([pre, n + pre, n + pre + 5000, 3 * n + pre, 3 * n + pre + 1, 100], dtype=np.int64)
    fp = FeatureProcessing(YAML, fd, CHANS, FS, engine="auto", nb_samples=n,
                           nb_pretrigger_samples=pre)
    df = fp.process_adc(adc, trig, scale, offset)
    lo = trig - pre
    ok = (lo >= 0) & (lo + n <= n_stream)
    assert list(ok) == [True, True, True, True, False, False]
    ev = np.zeros((len(trig), 4, n), dtype=np.float32)
    for b in np.nonzero(ok)[0]:
        for c in range(4):
            ev[b, c] = (adc[c, lo[b]:lo[b] + n].astype(np.float32) * np.float32(scale[c])
                        + np.float32(offset[c]))
    df2 = fp.process(ev, valid=ok.astype(np.uint8))
    assert list(df.columns) == list(df2.columns)
    assert np.array_equal(df.to_numpy(), df2.to_numpy())
    assert (df.iloc[4] == -999999.0).all() and (df.iloc[5] == -999999.0).all()
    # an algorithm with its own trace length gets its own window around the same trigger
    # (config.py:547-572, processing_data.py:640-656)
    n2, pre2 = 8192, 2048
    from detprocess_amd import OFPlan, build_filter
    t2 = synth.make_template(n2, pre2, FS)
    J2 = synth.make_psd(n2, FS)
    fd2, _ = _filter_data(n, pre)
    fd2.set_template("Melange1pc1ch", t2, sample_rate=FS, pretrigger_length_samples=pre2, tag="short")
    fd2.set_psd("Melange1pc1ch", J2, np.fft.fftfreq(n2, d=1 / FS), sample_rate=FS, tag="short")
    yaml2 = YAML.replace("    baseline:\n        run: True\n        window_min_from_start_usec: 0",
                         "    of1x1_short:\n        run: True\n        base_algorithm: of1x1_unconstrained\n"
                         "        template_tag: short\n        csd_tag: short\n"
                         f"        nb_samples: {n2}\n        nb_pretrigger_samples: {pre2}\n"
                         "    baseline:\n        run: True\n        window_min_from_start_usec: 0", 1)
    fp2 = FeatureProcessing(yaml2, fd2, CHANS, FS, engine="auto", nb_samples=n,
                            nb_pretrigger_samples=pre)
    df3 = fp2.process_adc(adc, trig, scale, offset)
    assert np.array_equal(df3["amp_of1x1_unconstrained_Melange1pc1ch"], df["amp_of1x1_unconstrained_Melange1pc1ch"])
    lo2 = trig - pre2
    ok2 = (lo2 >= 0) & (lo2 + n2 <= n_stream)
    ev2 = np.zeros((len(trig), n2), dtype=np.float32)
    for b in np.nonzero(ok2)[0]:
        ev2[b] = adc[0, lo2[b]:lo2[b] + n2].astype(np.float32) * np.float32(scale[0]) + np.float32(offset[0])
    p2 = OFPlan(n2, pre2, FS, max_batch=16)
    p2.set_filter(0, build_filter(t2, J2, FS, pre2))
    sid = p2.add_search(0, "delay")
    want = p2.process(ev2, valid=ok2.astype(np.uint8))
    assert np.array_equal(df3["amp_of1x1_short_Melange1pc1ch"].to_numpy(dtype=np.float32), want[:, 0])
    with pytest.raises(ValueError, match="Number of samples is not consistent"):
        fp2.process(ev)


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["fused", "rocfft", "lds"])
def test_psd_amp_and_energyabsorbed(engine):
    """psd_amp (algorithms.py:952-1044) and
    energyabsorbed (:889-949) reuse data the kernel already holds."""
    import torch
    from detprocess_amd import FeatureExtractors as FE, FeatureProcessing, OFBase, OFPlan, build_filter
    from oracle import of1x1 as orc
    n, pre = 32768, 16384
    fd, J = _filter_data(n, pre)
    tmpl = synth.make_template(n, pre, FS)
    ft = build_filter(tmpl, J, FS, pre)
    x, _, _ = synth.make_traces(7, tmpl, J, FS, ft.ampres, seed=77)
    x32 = (x + 2.5e-7).astype(np.float32)          # DC offset like a real TES current
    x64 = x32.astype(np.float64)
    f_lims = [[45.0, 75.0], [300.0, 500.0], [350.0, 450.0], [150, 250], [3000, 9000], 120.0]
    want = orc.psd_amp(x64, FS, f_lims)
    # C ABI level
    plan = OFPlan(n, pre, FS, max_batch=4, device=0, engine=engine)
    plan.set_filter(0, ft)
    plan.add_search(0, "nodelay")
    rng, names = utils.cleanup_freq_ranges(f_lims)
    ids = [plan.add_band(lo, hi) for lo, hi in utils.get_bin_ranges(rng, n, FS)]
    wb, ww = plan.add_tdwindow(0, 16000), plan.add_tdwindow(16000, 20000)
    out = plan.process(torch.as_tensor(x32, device="cuda:0")).cpu().numpy().astype(np.float64)
    for i, name in zip(ids, names):
        assert np.allclose(out[:, plan.band_offset(i)], want[name], rtol=1e-5), name
    o = plan.tdwindow_offset(ww)
    assert np.allclose(out[:, o + 4], x64[:, 16000:20000].sum(axis=1), rtol=2e-6)
    assert np.allclose(out[:, o + 5], (x64[:, 16000:20000] ** 2).sum(axis=1), rtol=2e-6)
    assert np.array_equal(out[:, o + 6], x64[:, 16000]) and np.array_equal(out[:, o + 7], x64[:, 19999])
    # static-method level
    vb, i0, rl = 190.6e-9, 88e-9, 8.8e-3
    e_ref = orc.energyabsorbed(x64, FS, vb, i0, rl, 16000, 20000)
    e = FE.energyabsorbed(x32, FS, vb, i0, rl, window_min_index=16000, window_max_index=20000)
    assert np.allclose(e["energyabsorbed"], e_ref, rtol=2e-4, atol=1e-6 * np.abs(e_ref).max())
    ob = OFBase(FS)
    ob.update_signal("A", x32)
    r = FE.psd_amp("A", ob, f_lims=f_lims)
    assert set(r) == {f"psd_amp_{nm}" for nm in names}
    for nm in names:
        assert np.allclose(r[f"psd_amp_{nm}"], want[nm], rtol=1e-5)
    with pytest.raises(ValueError):
        FE.psd_amp("A", ob)
    # batched driver with the reference's YAML keys
    yaml_txt = """
A:
    of1x1_nodelay:
        run: True
        template_tag: default
    psd_amp:
        run: True
        f_lims: [[45.0, 75.0], [300.0, 500.0], [3000, 9000]]
    energyabsorbed:
        run: True
        i0: 88e-9
        rl: 8.8e-3
        vb: 190.6e-9
        window_min_from_trig_usec: -100
        window_max_from_trig_usec: 1000
"""
    fdA = FilterData()
    fdA.set_template("A", tmpl, sample_rate=FS, pretrigger_length_samples=pre)
    fdA.set_psd("A", J, np.fft.fftfreq(n, d=1 / FS), sample_rate=FS)
    fp = FeatureProcessing(yaml_txt, fdA, ["A"], FS, nb_samples=n, nb_pretrigger_samples=pre,
                           engine=engine if engine == "rocfft" else "auto")
    df = fp.process(x32)
    assert np.allclose(df["psd_amp_45_75_A"], want["45_75"], rtol=1e-5)
    assert np.allclose(df["psd_amp_3000_9000_A"], want["3000_9000"], rtol=1e-5)
    lo, hi = orc.get_window_indices(n, pre, FS, window_min_from_trig_usec=-100,
                                    window_max_from_trig_usec=1000)
    e_ref = orc.energyabsorbed(x64, FS, 190.6e-9, 88e-9, 8.8e-3, lo, hi)
    assert np.allclose(df["energyabsorbed_A"], e_ref, rtol=2e-4, atol=1e-6 * np.abs(e_ref).max())


@pytest.mark.gpu
def test_command_line_driver(tmp_path):
    """scripts/process_features.py: YAML + filter file + events on disk -> numbered feature files."""
    import importlib.util
    import os
    import pandas as pd
    from detprocess_amd import FeatureProcessing
    from detprocess_amd.output import read_features
    n, pre, B = 32768, 16384, 10
    fd, J = _filter_data(n, pre)
    tmpl = synth.make_template(n, pre, FS)
    x, _, _ = synth.make_traces(B * 4, tmpl, J, FS, 1e-9, seed=8, max_delay=100)
    ev = x.reshape(B, 4, n).astype(np.float32)
    np.save(tmp_path / "events.npy", ev)
    fd.save_npz(str(tmp_path / "filter.npz"))
    yaml_text = YAML
    (tmp_path / "setup.yaml").write_text(yaml_text)
    spec = importlib.util.spec_from_file_location(
        "process_features", os.path.join(os.path.dirname(__file__), "..", "scripts",
                                         "process_features.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    files = mod.main(["--processing_setup", str(tmp_path / "setup.yaml"), "--filter_file",
                      str(tmp_path / "filter.npz"), "--events", str(tmp_path / "events.npy"),
                      "--channels", ",".join(CHANS), "--sample_rate", str(FS), "--save_path",
                      str(tmp_path / "out"), "--events_per_dump", "6", "--processing_id", "t1"])
    assert [os.path.basename(f)[-12:] for f in files] == ["_F0001.arrow", "_F0002.arrow"]
    assert os.path.basename(os.path.dirname(files[0])).startswith("t1_feature_I1_D")
    got = pd.concat([read_features(f) for f in files], ignore_index=True)
    want = FeatureProcessing(yaml_text, fd, CHANS, FS).process(ev)
    assert list(got["event_index"]) == list(range(B))
    for c in want.columns:
        assert np.array_equal(got[c].to_numpy(), want[c].to_numpy()), c


def test_window_indices_table_of_the_product_helper():
    """detprocess_amd.utils.get_window_indices (the host code the GPU plans are built from)
    against the hand-derived table of features.py:1243-1344: int() truncates toward zero BEFORE
    the pretrigger is added (-10 us -> 16372, not 16371), both ends clamp to [0, N-1], defaults
    are 0 and N-1, max < min raises."""
    from detprocess_amd import utils
    g = utils.get_window_indices
    N, pre = 32768, 16384
    assert g(N, pre, FS) == (0, N - 1)
    assert g(N, pre, FS, window_min_from_trig_usec=-10, window_max_from_trig_usec=10) == (16372, 16396)
    assert g(N, pre, FS, window_min_from_trig_usec=-400, window_max_from_trig_usec=400) == (15884, 16884)
    assert g(N, pre, FS, window_min_from_start_usec=100) == (125, N - 1)
    assert g(N, pre, FS, window_max_to_end_usec=100) == (0, N - 125 - 1)
    assert g(N, pre, FS, window_min_to_end_usec=1000, window_max_to_end_usec=100) == (N - 1251, N - 126)
    assert g(N, pre, FS, window_min_to_end_usec=-1000, window_max_to_end_usec=-100) == (N - 1251, N - 126)
    assert g(N, pre, FS, window_min_from_trig_usec=-1e9) == (0, N - 1)
    assert g(N, pre, FS, window_max_from_trig_usec=1e9) == (0, N - 1)
    assert g(N, pre, FS, window_min_from_start_usec=0, window_max_from_trig_usec=-2000) == (0, pre - 2500)
    assert g(N, pre, FS, window_min_from_trig_usec=2000, window_max_to_end_usec=0) == (pre + 2500, N - 1)
    assert g(25000, 12500, FS, window_min_from_trig_usec=-400, window_max_from_trig_usec=400) == (12000, 13000)
    # non-integer products truncate toward zero on both sides of the trigger
    assert g(N, pre, 1.0e6, window_min_from_trig_usec=-10.7, window_max_from_trig_usec=10.7) == (pre - 10, pre + 10)
    # precedence: from_start wins over to_end wins over from_trig (features.py:1305-1335)
    assert g(N, pre, FS, window_min_from_start_usec=8, window_min_from_trig_usec=-10) == (10, N - 1)
    assert g(N, pre, FS, unknown_key=3) == (0, N - 1)
    with pytest.raises(ValueError, match="max index smaller than min"):
        g(N, pre, FS, window_min_from_trig_usec=10, window_max_from_trig_usec=-10)
    assert utils.extract_window_indices is utils.get_window_indices


def test_duplicate_yaml_keys_are_refused():
    from detprocess_amd import YamlConfig
    bad = "A:\n    baseline:\n        run: True\n    baseline:\n        run: False\n"
    with pytest.raises(ValueError, match='Duplicate key "baseline"'):
        YamlConfig(bad, ["A"], sample_rate=FS)
    bad2 = "A:\n    baseline:\n        run: True\n        run: False\n"
    with pytest.raises(ValueError, match='Duplicate key "run"'):
        YamlConfig(bad2, ["A"], sample_rate=FS)
    # a merge key followed by an overriding key is legal YAML, not a duplicate
    merged = ("A:\n    baseline: &base\n        run: True\n        window_min_from_start_usec: 0\n"
              "        window_max_from_trig_usec: -100\n    baseline_late:\n        <<: *base\n"
              "        base_algorithm: baseline\n        window_max_from_trig_usec: -10\n")
    cfg = YamlConfig(merged, ["A"], sample_rate=FS).get_config()["feature"]["channels"]["A"]
    assert cfg["baseline_late"]["window_max_from_trig_usec"] == -10
    assert cfg["baseline_late"]["window_min_from_start_usec"] == 0 and cfg["baseline"]["window_max_from_trig_usec"] == -100
    # ... and a key written twice next to a merge key still is one
    bad3 = merged + "        window_max_from_trig_usec: -20\n"
    with pytest.raises(ValueError, match='Duplicate key "window_max_from_trig_usec"'):
        YamlConfig(bad3, ["A"], sample_rate=FS)


@pytest.mark.gpu
def test_combined_channel_needs_its_own_filter_entry_and_single_channels_ignore_weights():
    """processing_data.py:295, 345: templates and CSDs are looked up with the YAML channel
    expression itself ('A+B' needs an 'A+B' entry; no silent use of A's filter);
    processing_data.py:1033-1047: weights only act where there is a '+' / '-'."""
    from detprocess_amd import FeatureProcessing
    n, pre, B = 32768, 16384, 4
    fd = FilterData()
    f = np.fft.fftfreq(n, d=1 / FS)
    J = synth.make_psd(n, FS)
    for ch in ("A", "B"):
        fd.set_template(ch, synth.make_template(n, pre, FS), sample_rate=FS,
                        pretrigger_length_samples=pre, tag="default")
        fd.set_psd(ch, J, f, sample_rate=FS, tag="default")
    ev, _, _ = synth.make_traces(2 * B, synth.make_template(n, pre, FS), J, FS, 1e-9, seed=3)
    ev = ev.reshape(B, 2, n).astype(np.float32)
    summed = ("A+B:\n    weight_A: 0.9\n    weight_B: 1.1\n    of1x1_nodelay:\n        run: True\n"
              "        template_tag: default\n")
    with pytest.raises(ValueError, match='Channel "A\\+B" not available'):
        FeatureProcessing(summed, fd, ["A", "B"], FS).process(ev)
    single = ("A:\n    weight_A: 3.0\n    of1x1_nodelay:\n        run: True\n        template_tag: default\n"
              "    maximum:\n        run: True\n")
    plain = "A:\n    of1x1_nodelay:\n        run: True\n        template_tag: default\n    maximum:\n        run: True\n"
    a = FeatureProcessing(single, fd, ["A", "B"], FS).process(ev)
    b = FeatureProcessing(plain, fd, ["A", "B"], FS).process(ev)
    assert np.array_equal(a["amp_of1x1_nodelay_A"], b["amp_of1x1_nodelay_A"])
    assert np.array_equal(a["maximum_A"], ev[:, 0, :-1].max(axis=1).astype(np.float64))
    # a list / NumPy valid mask next to CUDA events (engine.py: torch.as_tensor on the mask)
    import torch
    fp = FeatureProcessing(plain, fd, ["A", "B"], FS)
    d1 = fp.process(torch.as_tensor(ev, device="cuda:0"), valid=[1, 0, 1, 1])
    assert (d1.iloc[1] == -999999.0).all() and np.array_equal(d1["maximum_A"][[0, 2, 3]], b["maximum_A"][[0, 2, 3]])
