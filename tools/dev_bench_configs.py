# dev: throughput of BASELINE configs[2] (constrained +-400us + integral + min/max fused) and configs[3]-like
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, synth_traces
N = 32768; fs = 1.25e6; pre = N // 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tmpl = synth.make_template(N, pre, fs); psd = synth.make_psd(N, fs)
ft = build_filter(tmpl, psd, fs, pre)
sigma = float(np.sqrt(np.median(psd) * fs))
traces, truth = synth_traces(B, N, tmpl, sigma, 3 * ft.ampres, 300 * ft.ampres, 0.5, 2000, seed=0)
def run(name, setup, x=traces):
    plan = OFPlan(N, pre, fs, max_batch=8192, engine='fused')
    setup(plan)
    out = plan.process(x); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): out = plan.process(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f'{name}: {x.shape[0]/dt/1e6:.3f} M events/s ({dt*1e3:.2f} ms) row={plan.row_floats}')
def c2(p): p.set_filter(0, ft); p.add_search(0, 'delay')
def c3(p):
    p.set_filter(0, ft); p.add_search(0, 'delay', 15884, 16884)
    p.add_tdwindow(pre - 12, pre + 625); p.add_tdwindow(pre - 625, pre + 625)
def c3b(p):
    p.set_filter(0, ft); p.add_search(0, 'delay', 15884, 16884)
def c3c(p):
    p.set_filter(0, ft); p.add_search(0, 'delay'); p.add_tdwindow(0, N - 1)
def c_all(p):
    p.set_filter(0, ft); p.add_search(0, 'nodelay'); p.add_search(0, 'delay'); p.add_search(0, 'delay', 15884, 16884)
    p.add_tdwindow(0, pre - 2500); p.add_tdwindow(pre + 2500, N - 1); p.add_tdwindow(pre - 625, pre + 625); p.add_tdwindow(pre - 12, pre + 625)
run('config2 unconstrained', c2)
run('config3 constrained + integral + min/max (2 windows)', c3)
run('constrained only', c3b)
run('unconstrained + 1 full td window', c3c)
run('3 searches + 4 td windows', c_all)
ev = traces[: B // 4 * 4].reshape(B // 4, 4, N)
tg = build_filter(synth.make_template(N, pre, fs, 'glitch'), psd, fs, pre)
tm = build_filter(synth.make_template(N, pre, fs, 'muon'), psd, fs, pre)
def c4(ch):
    def f(p):
        for s, t in enumerate((ft, tg, tm)):
            p.set_filter(s, t); p.add_search(s, 'nodelay'); p.add_search(s, 'delay'); p.add_search(s, 'delay', 15884, 16884)
        p.add_tdwindow(0, pre - 2500); p.add_tdwindow(pre - 625, pre + 625)
        p.set_channels(4, [ch], [1.0])
    return f
run('config4: one channel of 4, 3 template tags x 3 algos + 2 td windows', c4(0), ev)
