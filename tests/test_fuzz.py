"""A fixed slice of the randomised engine-vs-oracle cross-check (tools/fuzz_engines.py): random
trace lengths (also odd half-lengths and the 1024-thread LDS builds), pretrigger positions, batch
sizes, windows, outside-window and interpolated fits, one to three template tags, every engine
that accepts the case; and of the N x M engine (channel / template counts, channel maps, valid
masks, windows, every transform build) and of the trigger stage (filter and stream lengths around
the overlap-save block boundaries, 1 x 1 and N x M, padding on and off) and of the raw-data front end (windows cut on the GPU out of
int16 streams, also hanging over either end, against host-cut windows, bit for bit); and random
plans on the FUSED kernel (template tags, fit kinds, time-domain windows, channel sums, masks,
batches above the persistent grid) against the ROCFFT engine and the oracle."""
import importlib.util
import os

import pytest


@pytest.mark.gpu
def test_random_configurations_match_the_oracle():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "fuzz_engines.py")
    spec = importlib.util.spec_from_file_location("fuzz_engines", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(16, 2026, verbose=False) == 0
    assert mod.run_nxm(16, 2026, verbose=False) == 0
    assert mod.run_trigger(24, 2026, verbose=False) == 0
    assert mod.run_adc(10, 2026, verbose=False) == 0
    assert mod.run_fused(6, 2026, verbose=False) == 0
