# dev: ROCFFT-engine throughput for a given trace length (e.g. the reference example's 25000 samples)
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, synth_traces
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
fs = 1.25e6; pre = N // 2
tmpl = synth.make_template(N, pre, fs); psd = synth.make_psd(N, fs)
ft = build_filter(tmpl, psd, fs, pre)
sigma = float(np.sqrt(np.median(psd) * fs))
traces, truth = synth_traces(B, N, tmpl, sigma, 3 * ft.ampres, 300 * ft.ampres, 0.5, 2000, seed=0)
plan = OFPlan(N, pre, fs, max_batch=8192, engine='rocfft')
plan.set_filter(0, ft); plan.add_search(0, 'delay')
out = plan.process(traces); torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 3
for _ in range(reps): out = plan.process(traces)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f'rocfft N={N}: {B/dt/1e6:.3f} M traces/s ({dt*1e3:.2f} ms per {B}); {B/dt*N*4/1e12:.3f} TB/s algorithmic')
