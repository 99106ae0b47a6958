"""BASELINE configs[0]: 1k synthetic single-channel 4096-sample traces, of1x1_nodelay + baseline -- the
reference's own CPU-runnable case (FeatureProcessing._process per event, features.py:533-851, with
FeatureExtractors.of1x1_nodelay / baseline, algorithms.py:277-350, 650-700) -- through the two surfaces a
detprocess user touches: the YAML-driven batch driver and the FeatureExtractors static methods.  The GPU
side runs on the 4096-sample kernel (k_wave); the expected values are the fp64 oracle's, event by event."""
import numpy as np
import pytest

from detprocess_amd import FilterData, synth
from oracle import of1x1 as orc

FS = 1.25e6
N, PRE, B = 4096, 2048, 1000
CHAN = "Melange1pc1ch"
YAML0 = f"""
filter_file: /path/to/filter_file.hdf5
global:
    trace_length_msec: {N / FS * 1e3}
    pretrigger_length_msec: {PRE / FS * 1e3}
{CHAN}:
    of1x1_nodelay:
        run: True
        template_tag: default
    baseline:
        run: True
        window_min_from_start_usec: 0
        window_max_from_trig_usec: -800
"""


def _inputs():
    tmpl = synth.make_template(N, PRE, FS)
    J = synth.make_psd(N, FS)
    fd = FilterData()
    fd.set_template(CHAN, tmpl, sample_rate=FS, pretrigger_length_samples=PRE, tag="default")
    fd.set_psd(CHAN, J, np.fft.fftfreq(N, d=1 / FS), sample_rate=FS, tag="default")
    filt = orc.OFFilter(tmpl, J, FS, PRE)
    x, _, _ = synth.make_traces(B, tmpl, J, FS, filt.ampres, seed=2024, max_delay=0)
    return fd, tmpl, J, filt, x.astype(np.float32)


def test_the_oracle_on_configs0_recovers_the_injected_amplitudes():
    """CPU leg: the restated reference path on the 1k events -- no-delay amplitudes scatter around the
    injected ones with the filter's own resolution, chi2 around N - 1 degrees of freedom."""
    fd, tmpl, J, filt, x32 = _inputs()
    _, amps, _ = synth.make_traces(B, tmpl, J, FS, filt.ampres, seed=2024, max_delay=0)
    r = orc.process_events(filt, x32.astype(np.float64), "nodelay")
    pull = (r["amp"] - amps) / filt.ampres
    assert abs(pull.mean()) < 0.15 and 0.85 < pull.std() < 1.15
    assert abs(np.median(r["chi2"]) / (N - 1) - 1.0) < 0.05
    lo, hi = orc.get_window_indices(N, PRE, FS, window_min_from_start_usec=0, window_max_from_trig_usec=-800)
    assert (lo, hi) == (0, 1048)


@pytest.mark.gpu
def test_configs0_through_the_yaml_driver_and_the_static_methods():
    from detprocess_amd import FeatureExtractors as FE, FeatureProcessing, OFBase
    fd, tmpl, J, filt, x32 = _inputs()
    x64 = x32.astype(np.float64)
    ref = orc.process_events(filt, x64, "nodelay")
    lo, hi = orc.get_window_indices(N, PRE, FS, window_min_from_start_usec=0, window_max_from_trig_usec=-800)
    base = orc.baseline(x64, lo, hi)
    # 1. FeatureProcessing (YAML + filter data -> DataFrame), one batch
    fp = FeatureProcessing(YAML0, fd, [CHAN], FS)
    df = fp.process(x32.reshape(B, 1, N))
    assert all(p.engine == "fused" for p in fp.plans(N).values())      # the 4096-sample kernel
    assert f"t0_of1x1_nodelay_{CHAN}" not in df.columns                 # algorithms.py:344-348
    amp = df[f"amp_of1x1_nodelay_{CHAN}"].to_numpy()
    chi2 = df[f"chi2_of1x1_nodelay_{CHAN}"].to_numpy()
    low = df[f"lowchi2_of1x1_nodelay_{CHAN}"].to_numpy()
    chi0 = ref["chi2"] + ref["amp"] ** 2 * filt.norm          # chi2 = chi2_0 - A^2 norm (tests/util.py)
    assert np.all(np.abs(amp - ref["amp"]) <= 1e-5 * np.abs(ref["amp"]) + 1e-4 * filt.ampres)
    assert np.all(np.abs(chi2 - ref["chi2"]) <= 1e-5 * ref["chi2"] + 2e-6 * chi0)
    assert np.all(np.abs(low - ref["lowchi2"]) <= 1e-5 * ref["lowchi2"] + 2e-6 * chi0)
    sc = np.abs(x64).max()
    assert np.allclose(df[f"baseline_{CHAN}"].to_numpy(), base, rtol=1e-5, atol=2e-6 * sc)
    # 2. the static methods on an OFBase holding the batch (processing_data.py:731-772, then algorithms.py)
    ob = OFBase(FS)
    ob.set_csd(CHAN, J, coupling="AC")
    ob.add_template(CHAN, tmpl, template_tag="default", pretrigger_samples=PRE)
    ob.calc_phi(CHAN, "default")
    ob.update_signal(CHAN, x32)
    r = FE.of1x1_nodelay(CHAN, ob, template_tag="default")
    assert np.array_equal(r["amp_of1x1_nodelay"], amp) and np.array_equal(r["chi2_of1x1_nodelay"], chi2)
    b = FE.baseline(x32, lo, hi)["baseline"]
    assert np.allclose(b, base, rtol=1e-5, atol=2e-6 * sc)
