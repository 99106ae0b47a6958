# dev: throughput of the NxM engine (device-resident events), 2x2 and 3x2 at 32768 and 25000 samples
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from detprocess_amd.ofnxm import NxMPlan, build_nxm_filter, nxm_search_range
from test_ofnxm import make_csd, make_templates
FS = 1.25e6
def run(n, C, M, B, unconstrained=False):
    pre = n // 2
    t = make_templates(n, pre, C, M); csd = make_csd(n, C)
    plan = NxMPlan(build_nxm_filter(t, csd, FS, pre), max_batch=int(sys.argv[1]) if len(sys.argv) > 1 else 2048)
    plan.add_search('nodelay')
    lo, hi = nxm_search_range(n, pre, FS, -100, 100)
    plan.add_search('delay', *( (0, n) if unconstrained else (lo, hi)))
    ev = torch.randn(B, C, n, device='cuda') * 1e-9
    out = plan.process(ev); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): out = plan.process(ev)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    gb = B * n * (12 * C + 12 * M + (4 * M if unconstrained else 0)) / 1e9   # x r, Z w+r, Z' w+r, q w (+ r)
    print(f'N={n} {C}x{M} {"unconstrained" if unconstrained else "constrained"}: {B/dt/1e6:.3f} M events/s '
          f'({dt*1e3:.1f} ms, >= {gb/dt:.0f} GB/s of pass traffic)', flush=True)
run(32768, 2, 2, 8192)
run(32768, 2, 2, 8192, True)
run(32768, 3, 2, 8192)
run(25000, 2, 2, 8192)
run(4096, 2, 2, 65536)
