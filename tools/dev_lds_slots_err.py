import sys
import numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth
from oracle import of1x1 as orc
FS = 1.25e6
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
pre = n // 2
psd = synth.make_psd(n, FS)
kinds = ("pulse", "glitch", "muon")
tmpls = [synth.make_template(n, pre, FS, k) for k in kinds]
fts = [build_filter(t, psd, FS, pre) for t in tmpls]
filts = [orc.OFFilter(t, psd, FS, pre) for t in tmpls]
x, _, _ = synth.make_traces(21, tmpls[0], psd, FS, fts[0].ampres, seed=314)
x32 = x.astype(np.float32); x64 = x32.astype(np.float64)
for nslots in (3, 1):
    for s0 in range(3 if nslots == 1 else 1):
        plan = OFPlan(n, pre, FS, max_batch=64, device=0, engine='lds')
        use = list(range(3)) if nslots == 3 else [s0]
        ids = {}
        for j, s in enumerate(use):
            plan.set_filter(j, fts[s]); ids[s] = (j, plan.add_search(j, 'nodelay'), plan.add_search(j, 'delay'))
        out = plan.process(torch.as_tensor(x32, device='cuda')).cpu().numpy().astype(np.float64)
        for s in use:
            j, a, b = ids[s]
            r = orc.process_events(filts[s], x64, 'nodelay'); o = plan.search_offset(j, a)
            e = np.abs(out[:, o] - r['amp']) / fts[s].ampres
            r2 = orc.process_events(filts[s], x64, 'unconstrained'); o2 = plan.search_offset(j, b)
            e2 = np.abs(out[:, o2] - r2['amp']) / fts[s].ampres
            print(f'N={n} slots={nslots} {kinds[s]}: nodelay err/sigma max {e.max():.2e}  delay err/sigma {e2.max():.2e} bins ok {np.array_equal(out[:, o2+7], r2["index"])}', flush=True)
        plan.close()
