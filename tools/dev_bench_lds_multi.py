# dev: LDS engine with three template slots at the reference example's trace length
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, synth_traces
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
fs = 1.25e6; pre = N // 2; B = 65536
psd = synth.make_psd(N, fs)
fts = [build_filter(synth.make_template(N, pre, fs, k), psd, fs, pre) for k in ('pulse', 'glitch', 'muon')]
traces, _ = synth_traces(B, N, synth.make_template(N, pre, fs), 1e-9, 3 * fts[0].ampres, 300 * fts[0].ampres, 0.5, 2000, seed=0)
for nsl in (1, 3):
    plan = OFPlan(N, pre, fs, max_batch=8192, engine='lds')
    for s in range(nsl):
        plan.set_filter(s, fts[s]); plan.add_search(s, 'delay')
    out = plan.process(traces); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out = plan.process(traces)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'lds N={N} slots={nsl}: {B/dt/1e6:.3f} M events/s')
