import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from detprocess_amd.ofnxm import NxMPlan, build_nxm_filter
from oracle import ofnxm as onm
from test_ofnxm import make_csd, make_templates, make_events, FS
for (n, pre, C, M) in [(4096, 2048, 2, 2), (32768, 16384, 2, 2), (25000, 12500, 3, 2), (2048, 500, 1, 1), (4096, 2048, 4, 4), (8192, 4096, 2, 3)]:
    t = make_templates(n, pre, C, M); csd = make_csd(n, C)
    filt = onm.NxMFilter(t, csd, FS, pre)
    B = 24 if n <= 8192 else 10
    ev, _, _ = make_events(B, t, csd, filt.ampres, seed=n + C, max_delay=min(n // 8, 2000))
    plan = NxMPlan(build_nxm_filter(t, csd, FS, pre), max_batch=16)
    s0 = plan.add_search('nodelay'); s1 = plan.add_search('delay')
    out = plan.process(ev.astype(np.float32)).astype(np.float64)
    ref = onm.process_events(filt, ev.astype(np.float32).astype(np.float64))
    a0, _, c0, _ = plan.record(out, s0); a1, t1, c1, i1 = plan.record(out, s1)
    e0 = np.abs(a0 - ref['amps_nodelay']); e1 = np.abs(a1 - ref['amps'])
    print(f'N={n} {C}x{M} cond(P)={np.linalg.cond(filt.P):.1f}: nodelay err/sigma max {np.max(e0/filt.ampres):.2e}, rel max {np.max(e0/np.maximum(np.abs(ref["amps_nodelay"]),filt.ampres)):.2e}; '
          f'delay err/sigma {np.max(e1/filt.ampres):.2e} rel {np.max(e1/np.maximum(np.abs(ref["amps"]),filt.ampres)):.2e}; bins equal {np.array_equal(i1, ref["index"])}; chi2 rel {np.max(np.abs(c1-ref["chi2"])/ref["chi2"]):.2e} chi0-rel {np.max(np.abs(c1-ref["chi2"])/ref["chi2_0"]):.2e}', flush=True)
