# dev scratch: quick parity check of the ROCFFT engine vs oracle (run on the GPU box)
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth
from oracle import of1x1 as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = sys.argv[2] if len(sys.argv) > 2 else 'rocfft'
B = 64
fs = 1.25e6; pre = N // 2
tmpl = synth.make_template(N, pre, fs); psd = synth.make_psd(N, fs)
ft = build_filter(tmpl, psd, fs, pre)
of = orc.OFFilter(tmpl, psd, fs, pre)
print('norm', ft.norm, of.norm, 'ampres', ft.ampres)
traces, amps, delays = synth.make_traces(B, tmpl, psd, fs, ft.ampres, seed=1, max_delay=N//8)
tr32 = traces.astype(np.float32)
plan = OFPlan(N, pre, fs, max_batch=32, engine=eng)
plan.set_filter(0, ft)
s_un = plan.add_search(0, 'delay')
s_nd = plan.add_search(0, 'nodelay')
lo, hi = pre - 500*N//32768, pre + 500*N//32768
s_c = plan.add_search(0, 'delay', lo, hi)
w0 = plan.add_tdwindow(0, N - 1)
w1 = plan.add_tdwindow(100, pre)
print('row', plan.row_floats, plan.engine)
out = plan.process(torch.as_tensor(tr32, device='cuda')).cpu().numpy().astype(np.float64)
out_h = plan.process(tr32)
print('host-path equal:', np.array_equal(out_h.astype(np.float64), out))
ref = orc.process_events(of, tr32.astype(np.float64), 'unconstrained')
def cmp(name, a, b, scale=None):
    d = np.abs(a - b); s = np.abs(b) if scale is None else scale
    print(f'{name:24s} max abs {d.max():.3e} max rel {np.max(d/np.maximum(s,1e-300)):.3e}')
o = plan.search_offset(0, s_un)
cmp('amp', out[:, o+0], ref['amp'], ft.ampres)
print('t0 idx mismatches', np.sum(out[:, o+7] != ref['index']))
cmp('t0', out[:, o+1], ref['t0'], 1/fs)
cmp('chi2', out[:, o+2], ref['chi2'])
cmp('lowchi2', out[:, o+3], ref['lowchi2'])
cmp('chi0', out[:, o+4], ref['chi2nopulse'])
cmp('timeres', out[:, o+6], ref['timeres'])
refn = orc.process_events(of, tr32.astype(np.float64), 'nodelay')
o = plan.search_offset(0, s_nd)
cmp('nd amp', out[:, o+0], refn['amp'], ft.ampres)
cmp('nd chi2', out[:, o+2], refn['chi2'])
cmp('nd lowchi2', out[:, o+3], refn['lowchi2'])
refc = orc.process_events(of, tr32.astype(np.float64), 'constrained', window_min_index=lo, window_max_index=hi)
o = plan.search_offset(0, s_c)
cmp('c amp', out[:, o+0], refc['amp'], ft.ampres)
print('c idx mismatches', np.sum(out[:, o+7] != refc['index']))
cmp('c chi2', out[:, o+2], refc['chi2'])
x = tr32.astype(np.float64)
o = plan.tdwindow_offset(w0)
rms = x.std()
cmp('baseline', out[:, o+0], orc.baseline(x, 0, N-1), rms)
cmp('integral', out[:, o+1], orc.integral(x, fs, 0, N-1), rms*N/fs)
cmp('max', out[:, o+2], orc.maximum(x, 0, N-1))
cmp('min', out[:, o+3], orc.minimum(x, 0, N-1))
o = plan.tdwindow_offset(w1)
cmp('baseline w1', out[:, o+0], orc.baseline(x, 100, pre), rms)
