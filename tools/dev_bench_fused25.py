"""Development bench of the 25000-sample FUSED kernel against the LDS engine (same inputs)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from detprocess_amd import OFPlan, build_filter, synth, synth_traces

FS, N = 1.25e6, int(os.environ.get('N', 25000))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
pre = N // 2
tmpl = synth.make_template(N, pre, FS); psd = synth.make_psd(N, FS)
ft = build_filter(tmpl, psd, FS, pre)
x, _ = synth_traces(B, N, tmpl, 0.0, 30 * ft.ampres, 300 * ft.ampres, 0.5, 2000, seed=1, psd=psd, fs=FS)
for engine in sys.argv[2:] or ("fused", "lds"):
    p = OFPlan(N, pre, FS, max_batch=8192, device=0, engine=engine)
    p.set_filter(0, ft); p.add_search(0, "delay")
    out = torch.empty((B, p.row_floats), dtype=torch.float32, device="cuda:0")
    for _ in range(2): p.process(x, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): p.process(x, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{engine}: {B / dt / 1e6:.2f} M traces/s  ({dt * 1e3:.2f} ms, {B * (N * 4 + 16) / dt / 8e12:.3f} of the HBM roofline)", flush=True)
