import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, synth_traces
N = 32768; fs = 1.25e6; pre = N // 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
engines = sys.argv[2].split(',') if len(sys.argv) > 2 else ['fused', 'rocfft']
tmpl = synth.make_template(N, pre, fs); psd = synth.make_psd(N, fs)
ft = build_filter(tmpl, psd, fs, pre)
sigma = float(np.sqrt(np.median(psd) * fs))
traces, truth = synth_traces(B, N, tmpl, sigma, 3 * ft.ampres, 300 * ft.ampres, 0.5, 2000, seed=0)
torch.cuda.synchronize()
res = {}
for eng in engines:
    plan = OFPlan(N, pre, fs, max_batch=8192, engine=eng)
    plan.set_filter(0, ft)
    plan.add_search(0, 'delay')
    out = plan.process(traces); torch.cuda.synchronize()
    plan.enable_timing(True)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = plan.process(traces)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ms, nl = plan.kernel_time()
    print(f'{eng}: {B/dt/1e6:.3f} M traces/s  ({dt*1e3:.2f} ms per {B}); kernel avg {ms:.3f} ms x {nl}; HBM-roofline frac {B/dt*131088/8e12:.4f}')
    res[eng] = out.cpu().numpy()
if len(res) == 2:
    a, b = res['fused'], res['rocfft']
    print('idx mismatch fused vs rocfft:', int((a[:, 7] != b[:, 7]).sum()), 'of', B)
    print('amp max rel diff:', float(np.max(np.abs(a[:, 0] - b[:, 0])) / ft.ampres))
    tr = truth.cpu().numpy()
    has = tr[:, 0] > 10 * ft.ampres
    print('truth check (amp>10 sigma): delay match', float(np.mean((a[has, 7] - pre) == tr[has, 1])), 'amp rel err med', float(np.median(np.abs(a[has, 0] / tr[has, 0] - 1))))
