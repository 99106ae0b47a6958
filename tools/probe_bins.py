"""Black-box probe of the FUSED kernel at 25000 samples: one event per frequency bin (a pure cosine),
so that chi2_0 = 2 g_k |X_k|^2 isolates the weight the kernel applies to bin k.  It is what located
round 2's wrong k_fused25<2, true> / <6, true>: only the bins of thread 0's two
self-paired blocks were wrong.  OFX_LIB=<variant .so> CH=<channels> python tools/probe_bins.py"""
import os, sys
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, '..'))
import numpy as np, torch
from detprocess_amd import OFPlan, build_filter, synth
from oracle import of1x1 as orc
FS, N = 1.25e6, 25000
M = N // 2
psd = synth.make_psd(N, FS)
pre = N // 2
kinds = ['pulse', 'glitch']
tm = [synth.make_template(N, pre, FS, k) for k in kinds]
fts = [build_filter(t, psd, FS, pre) for t in tm]
filt = [orc.OFFilter(t, psd, FS, pre) for t in tm]
T = [(N // 5, (4 * N) // 5), (N // 2 + 1000, N // 2 + 5000)]
CH = int(os.environ.get('CH', '1'))
p = OFPlan(N, pre, FS, max_batch=4096, engine='fused')
for s, ft in enumerate(fts):
    p.set_filter(s, ft)
    if s == 1: p.add_search(s, 'nodelay', 0, N, False, 50000.0)
    p.add_search(s, 'delay', 0, N, False, 50000.0)
for lo, hi in T: p.add_tdwindow(lo, hi)
if CH > 1: p.set_channels(CH, [CH - 1], [1.0])
ks = np.arange(1, M)                 # bins 1 .. M-1
n = np.arange(N)
bad = {0: [], 1: []}
for c0 in range(0, len(ks), 2048):
    kk = ks[c0:c0 + 2048]
    x = (1e-8 * np.cos(2 * np.pi * np.outer(kk, n) / N + 0.3)).astype(np.float32)
    if CH > 1:
        xe = np.zeros((x.shape[0], CH, N), dtype=np.float32); xe[:, CH - 1] = x
        out = p.process(torch.as_tensor(xe, device='cuda')).cpu().numpy().astype(np.float64)
    else:
        out = p.process(torch.as_tensor(x, device='cuda')).cpu().numpy().astype(np.float64)
    V = np.fft.rfft(x.astype(np.float64), axis=1)
    for s in range(2):
        g = filt[s].g[:M + 1]
        w = np.full(M + 1, 2.0); w[0] = 1.0; w[M] = 1.0
        chi0 = (w * g * (V.real ** 2 + V.imag ** 2)).sum(axis=1)
        o = p.search_offset(s, 1 if s == 1 else 0)
        rel = (out[:, o + 4] - chi0) / chi0
        for i in np.nonzero(np.abs(rel) > 1e-4)[0]:
            bad[s].append((int(kk[i]), float(rel[i])))
for s in range(2):
    print(f'slot {s}: {len(bad[s])} bins with wrong chi2_0 weight')
    rows = []
    for k, r in bad[s]:
        # bin k sits in thread v = k mod 500 (or 500 - that), slot J = k // 500 ... (k = v + 500 J or p = M - k)
        v = k % 500; J = k // 500
        if v > 250: v2, J2, side = 500 - v, 24 - J, 'p'
        else: v2, J2, side = v, J, 'k'
        rows.append((k, v2, J2, side, float('%.3g' % r)))
    print('   (bin, thread, slot J, side, rel err):', rows[:12])
    if rows:
        th = sorted({r[1] for r in rows}); Js = sorted({(r[2], r[3]) for r in rows})
        print('   threads:', th[:12], ' n =', len(th)); print('   slots:', Js[:14])
p.close()
