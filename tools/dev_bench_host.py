# dev: PCIe-inclusive throughput of ofx_process on host (NumPy) buffers, float32 and int16-stream paths
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth
N = 32768; fs = 1.25e6; pre = N // 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
tmpl = synth.make_template(N, pre, fs); psd = synth.make_psd(N, fs)
ft = build_filter(tmpl, psd, fs, pre)
rng = np.random.default_rng(0)
x = rng.standard_normal((B, N), dtype=np.float32) * 1e-9
plan = OFPlan(N, pre, fs, max_batch=4096, engine='fused')
plan.set_filter(0, ft); plan.add_search(0, 'delay')
out = plan.process(x[:4096])
t0 = time.perf_counter(); out = plan.process(x); dt = time.perf_counter() - t0
print(f'float32 host events: {B/dt/1e3:.1f} k events/s ({B*N*4/dt/1e9:.1f} GB/s over PCIe)')
adc = rng.integers(-2000, 2000, size=(1, B * N // 4), dtype=np.int16)
trig = np.sort(rng.integers(pre, adc.shape[1] - N, B)).astype(np.int64)
o = plan.process_adc(adc[:, :N * 8], trig[:8] % (N * 4) + pre, 1e-12, 0.0)
t0 = time.perf_counter(); o = plan.process_adc(adc, trig, 1e-12, 0.0); dt = time.perf_counter() - t0
print(f'int16 stream, {B} overlapping windows from {adc.shape[1]/1e6:.0f} M samples: {B/dt/1e3:.1f} k events/s')
