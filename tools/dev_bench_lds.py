# dev: LDS-engine throughput for a given trace length, next to the ROCFFT engine
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, synth_traces
fs = 1.25e6
for N in [int(a) for a in sys.argv[1:]] or [4096, 25000]:
    pre = N // 2; B = 2147483648 // N // 4
    tmpl = synth.make_template(N, pre, fs); psd = synth.make_psd(N, fs); ft = build_filter(tmpl, psd, fs, pre)
    sigma = float(np.sqrt(np.median(psd) * fs))
    traces, _ = synth_traces(B, N, tmpl, sigma, 3 * ft.ampres, 300 * ft.ampres, 0.5, min(2000, N // 8), seed=0)
    for eng in ('lds', 'rocfft'):
        plan = OFPlan(N, pre, fs, max_batch=8192, engine=eng); plan.set_filter(0, ft); plan.add_search(0, 'delay')
        out = plan.process(traces); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): out = plan.process(traces)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f'{eng:7s} N={N}: {B/dt/1e6:8.3f} M traces/s; {B/dt*N*4/1e12:.3f} TB/s algorithmic')
