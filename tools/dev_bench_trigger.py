# dev: throughput of the GPU trigger on one minute of 1.25 MHz data (75 M samples), vs the oracle on a slice
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OptimumFilterTrigger, synth
from oracle import oftrigger as ot
fs = 1.25e6; N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768; pre = N // 2
L = int(sys.argv[2]) if len(sys.argv) > 2 else 75_000_000
tmpl = synth.make_template(N, pre, fs); psd = synth.make_psd(N, fs)
g = OptimumFilterTrigger('chanA', fs, tmpl, psd, pre)
rng = np.random.default_rng(0)
x = torch.randn(L, device='cuda') * float(np.sqrt(np.median(psd) * fs / 2))
for p in rng.integers(N, L - 2 * N, 2000):
    x[p:p + N] += torch.as_tensor(tmpl * 20 * g.get_resolution()[0], device='cuda', dtype=torch.float32)
g.update_trace(x); g.find_triggers(5.0, pileup_window_msec=1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): g.update_trace(x)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(3): g.find_triggers(5.0, pileup_window_msec=1.0)
t2 = time.perf_counter()
n = len(g.get_trigger_data()['chanA']['trigger_index'])
print(f'GPU trigger N={N}: update_trace {(t1-t0)/3*1e3:.1f} ms ({L/((t1-t0)/3)/1e9:.2f} Gsamples/s), find_triggers {(t2-t1)/3*1e3:.1f} ms, {n} triggers')
xs = x[:3_000_000].cpu().numpy().astype(np.float64)
o = ot.OFTrigger(fs, tmpl, psd, pre)
t0 = time.perf_counter(); o.update_trace(xs); o.find_triggers(5.0, pileup_window_msec=1.0); dt = time.perf_counter() - t0
print(f'CPU oracle (scipy oaconvolve, 1 process): {len(xs)/dt/1e6:.1f} Msamples/s')
