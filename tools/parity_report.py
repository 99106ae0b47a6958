# Parity report asked for by SURVEY.md section 8(d): the three engines against the fp64 oracle on the
# same seeded events (the bench recipe: coloured noise drawn from J, half of the events with a pulse):
# max / 99.9-percentile relative error of amp, chi2, lowchi2, exact-match rate of the t0 bin and the
# near-tie flips.  Usage (GPU box):  python tools/parity_report.py [n_events] [out.json]
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from concurrent.futures import ProcessPoolExecutor
from detprocess_amd import OFPlan, build_filter, synth
from oracle import of1x1 as orc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dest = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/parity_report.json'
FS = 1.25e6


def _oracle_chunk(args):
    n, pre, x = args
    tmpl = synth.make_template(n, pre, FS); psd = synth.make_psd(n, FS)
    filt = orc.OFFilter(tmpl, psd, FS, pre)
    return orc.process_events(filt, x.astype(np.float64), 'unconstrained')


def report(n, engines, b):
    pre = n // 2
    tmpl = synth.make_template(n, pre, FS); psd = synth.make_psd(n, FS)
    ft = build_filter(tmpl, psd, FS, pre)
    x, _, _ = synth.make_traces(b, tmpl, psd, FS, ft.ampres, seed=11, max_delay=min(2000, n // 8))
    x = x.astype(np.float32)
    t0 = time.time()
    parts = np.array_split(np.arange(b), 16)
    with ProcessPoolExecutor(8) as ex:
        res = list(ex.map(_oracle_chunk, [(n, pre, x[p]) for p in parts]))
    ref = {k: np.concatenate([r[k] for r in res]) for k in res[0]}
    print(f'N={n}: oracle on {b} events in {time.time() - t0:.0f} s', flush=True)
    out = {}
    for eng in engines:
        plan = OFPlan(n, pre, FS, max_batch=4096, engine=eng)
        plan.set_filter(0, ft); sid = plan.add_search(0, 'delay'); o = plan.search_offset(0, sid)
        g = plan.process(torch.as_tensor(x, device='cuda')).cpu().numpy().astype(np.float64)
        same = g[:, o + 7].astype(np.int64) == ref['index']
        rel = {'amp': np.abs(g[:, o] - ref['amp']) / np.maximum(np.abs(ref['amp']), ft.ampres),
               'chi2': np.abs(g[:, o + 2] - ref['chi2']) / np.abs(ref['chi2']),
               'lowchi2': np.abs(g[:, o + 3] - ref['lowchi2']) / np.abs(ref['lowchi2'])}
        flips = np.nonzero(~same)[0]
        r = {'events': int(b), 't0_bin_exact_rate': float(same.mean()), 't0_bin_flips': int((~same).sum())}
        for k, v in rel.items():
            v = v[same]
            r[f'{k}_max_rel'] = float(v.max()); r[f'{k}_p999_rel'] = float(np.quantile(v, 0.999))
        if flips.size:
            # chi2 difference (relative to chi2_0) between the two candidate bins, from the engine itself
            d = np.abs(g[flips, o + 2] - ref['chi2'][flips]) / ref['chi2nopulse'][flips]
            r['flip_chi2_gap_rel_chi2_0_max'] = float(d.max())
        out[eng] = r
        print(eng, json.dumps(r), flush=True)
        plan.close()
    return out


if __name__ == '__main__':
    rep = {'recipe': 'synth.make_traces seed 11 (coloured noise from J, 50 % pulses 3-300 sigma, |d| <= 2000), '
                     'of1x1_unconstrained, relative errors over the events whose t0 bin matches; amp relative to '
                     'max(|amp|, ampres)',
           'n32768': report(32768, ('fused', 'rocfft'), B),
           'n25000': report(25000, ('fused', 'lds', 'rocfft'), B // 2),
           'n4096': report(4096, ('fused', 'lds', 'rocfft'), B)}
    json.dump(rep, open(dest, 'w'), indent=1)
    print('written', dest)
