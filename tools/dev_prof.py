import sys
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, synth_traces
N = 32768; fs = 1.25e6; pre = N // 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
eng = sys.argv[2] if len(sys.argv) > 2 else 'fused'
tmpl = synth.make_template(N, pre, fs); psd = synth.make_psd(N, fs)
ft = build_filter(tmpl, psd, fs, pre)
sigma = float(np.sqrt(np.median(psd) * fs))
traces, truth = synth_traces(B, N, tmpl, sigma, 3 * ft.ampres, 300 * ft.ampres, 0.5, 2000, seed=0)
plan = OFPlan(N, pre, fs, max_batch=8192, engine=eng)
plan.set_filter(0, ft)
plan.add_search(0, 'delay')
for _ in range(3):
    out = plan.process(traces)
torch.cuda.synchronize()
print('done', float(out[:, 0].abs().mean()))
