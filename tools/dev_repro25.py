"""Development: isolate a k_fused25 discrepancy found by the fuzz (seed 41, case 4)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from detprocess_amd import OFPlan, build_filter, synth
from oracle import of1x1 as orc
FS, N = 1.25e6, int(os.environ.get('N', 25000))
psd = synth.make_psd(N, FS)


def run(pre, nslots, tdw, n_total, chan, B, fcut=50000.0, eng='fused', mb=64):
    kinds = ['pulse', 'glitch', 'muon'][:nslots]
    tm = [synth.make_template(N, pre, FS, k) for k in kinds]
    fts = [build_filter(t, psd, FS, pre) for t in tm]
    x, _, _ = synth.make_traces(B * n_total, tm[0], psd, FS, fts[0].ampres, seed=7, max_delay=2000)
    ev = x.reshape(B, n_total, N).astype(np.float32)
    p = OFPlan(N, pre, FS, max_batch=mb, engine=eng)
    for s, ft in enumerate(fts):
        p.set_filter(s, ft)
        if s == 1: p.add_search(s, 'nodelay', 0, N, False, fcut)
        p.add_search(s, 'delay', 0, N, False, fcut)
    for lo, hi in tdw: p.add_tdwindow(lo, hi)
    if n_total > 1: p.set_channels(n_total, [chan], [1.0])
    inp = torch.as_tensor(ev if n_total > 1 else ev[:, 0], device='cuda')
    out = p.process(inp).cpu().numpy().astype(np.float64)
    ref = orc.process_events(orc.OFFilter(tm[0], psd, FS, pre), ev[:, chan].astype(np.float64), 'unconstrained', fcut)
    o = p.search_offset(0, 0)
    bad = np.nonzero(out[:, o + 7].astype(int) != ref['index'])[0]
    rel = np.abs(out[:, o + 2] - ref['chi2']) / ref['chi2']
    w = int(rel.argmax())
    print('   worst event', w, 'chi0 rel', abs(out[w, o + 4] - ref['chi2nopulse'][w]) / ref['chi2nopulse'][w], 'amp', out[w, o], ref['amp'][w],
          'idx', out[w, o + 7], ref['index'][w], 'low', out[w, o + 3], ref['lowchi2'][w], ' n(rel>1e-5)', int((rel > 1e-5).sum()))
    print(f'pre={pre} slots={nslots} td={len(tdw)} chans={n_total} B={B} mb={mb}: flips {len(bad)} first {bad[:6]}  chi2 max rel {rel.max():.2e} at {rel.argmax()}', flush=True)
    p.close()


T = [(N // 5, (4 * N) // 5), (N // 2 + 1000, N // 2 + 5000)]
P = N // 2
run(P, 2, T, 3, 2, 300)
