"""Development bench: the per-channel feature block of the reference's example YAML
(examples/processing/process_example.yaml:109-163: one template tag, of1x1_nodelay / _unconstrained /
_constrained +-100 us with lowchi2_fcutoff 50 kHz, five time-domain windows, five psd_amp bands) at its own
25000 samples, built up feature by feature on one plan, to see what each part costs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from detprocess_amd import OFPlan, build_filter, synth, synth_traces, search_range, utils

FS = 1.25e6
N = int(os.environ.get('N', 25000))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
eng = sys.argv[2] if len(sys.argv) > 2 else 'fused'
pre = N // 2
tmpl = synth.make_template(N, pre, FS); psd = synth.make_psd(N, FS)
ft = build_filter(tmpl, psd, FS, pre)
x, _ = synth_traces(B, N, tmpl, 0.0, 30 * ft.ampres, 300 * ft.ampres, 0.5, 2000, seed=1, psd=psd, fs=FS)
W = lambda **kw: utils.get_window_indices(nb_samples=N, nb_pretrigger_samples=pre, fs=FS, **kw)
wins = [W(window_min_from_start_usec=0, window_max_from_trig_usec=-2000),
        W(window_min_from_trig_usec=2000, window_max_to_end_usec=0),
        W(window_min_from_trig_usec=-500, window_max_from_trig_usec=500),
        W(window_min_from_trig_usec=-500, window_max_from_trig_usec=500),
        W(window_min_from_trig_usec=-10, window_max_from_trig_usec=500)]
df = FS / N
bands = [(max(1, int(a / df)), int(b / df) + 1) for a, b in ((45, 75), (300, 500), (350, 450), (150, 250), (250, 350))]
lo, hi = search_range(N, pre, FS, -100, 100)


def run(tag, build):
    p = OFPlan(N, pre, FS, max_batch=8192, device=0, engine=eng)
    p.set_filter(0, ft)
    build(p)
    out = torch.empty((B, p.row_floats), dtype=torch.float32, device='cuda:0')
    for _ in range(2): p.process(x, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): p.process(x, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    print(f'{tag:58s} {B / dt / 1e6:6.2f} M traces/s', flush=True)
    p.close()


def s_unc(p, fc=10000.0): p.add_search(0, 'delay', lowchi2_fcutoff=fc)
def s_all(p, fc=10000.0):
    p.add_search(0, 'nodelay', lowchi2_fcutoff=fc); s_unc(p, fc); p.add_search(0, 'delay', lo, hi, lowchi2_fcutoff=fc)
run('unconstrained, 10 kHz', s_unc)
run('unconstrained, 50 kHz', lambda p: s_unc(p, 50000.0))
run('nodelay + unconstrained + constrained, 10 kHz', s_all)
run('nodelay + unconstrained + constrained, 50 kHz', lambda p: s_all(p, 50000.0))
run('unconstrained + 5 windows', lambda p: (s_unc(p), [p.add_tdwindow(*w) for w in wins]))
run('unconstrained + 5 bands', lambda p: (s_unc(p), [p.add_band(*b) for b in bands]))
run('the example block (3 fits at 50 kHz, 5 windows, 5 bands)',
    lambda p: (s_all(p, 50000.0), [p.add_tdwindow(*w) for w in wins], [p.add_band(*b) for b in bands]))
