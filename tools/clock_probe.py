"""Run the FUSED kernel back-to-back for a few seconds and sample sclk / power with rocm-smi.
usage: python tools/clock_probe.py [n_traces] [seconds]"""
import subprocess, sys, threading, time, re
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, synth_traces
N = 32768; fs = 1.25e6; pre = N // 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
tmpl = synth.make_template(N, pre, fs); psd = synth.make_psd(N, fs)
ft = build_filter(tmpl, psd, fs, pre)
sigma = float(np.sqrt(np.median(psd) * fs))
traces, _ = synth_traces(B, N, tmpl, sigma, 3 * ft.ampres, 300 * ft.ampres, 0.5, 2000, seed=0)
plan = OFPlan(N, pre, fs, max_batch=8192, engine='fused')
plan.set_filter(0, ft); plan.add_search(0, 'delay')
out = plan.process(traces); torch.cuda.synchronize()
stop = False; samples = []
def watch():
    while not stop:
        t = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
        s = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', t); p = re.search(r'Power \(W\): ([\d.]+)', t)
        samples.append((int(s.group(1)) if s else -1, float(p.group(1)) if p else -1))
th = threading.Thread(target=watch); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(10): plan.process(traces, out=out)
    torch.cuda.synchronize(); n += 10
dt = time.time() - t0
stop = True; th.join()
print(f'{n * B / dt / 1e6:.2f} M traces/s over {dt:.1f} s')
print('sclk MHz / W samples:', samples)
