# Randomised cross-check of the engines against the fp64 oracle on small random configurations
# (trace length, pretrigger, batch size, windows, outside-window, interpolation, several slots).
# usage (GPU box): python tools/fuzz_engines.py [n_cases] [seed]
import sys, traceback
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from detprocess_amd import OFPlan, build_filter, synth
from detprocess_amd import _lib
from oracle import of1x1 as orc
from util import check_search

FS = 1.25e6


def run(cases, seed, verbose=True):
  rng = np.random.default_rng(seed)
  bad = 0
  for c in range(cases):
      n = int(rng.choice([2 * int(rng.integers(128, 3000)), int(rng.choice([256, 500, 1000, 1024, 2000, 2500, 3000, 4096, 5000,
                                                                              6000, 6250, 8192, 10000, 12000, 12500, 16384, 20000, 24000,
                                                                              25000, 30000, 32768]))]))
      pre = int(rng.integers(n // 8, n - n // 8))
      B = int(rng.integers(1, 70)) if n <= 12500 else int(rng.integers(1, 9))
      kinds = ['pulse', 'glitch', 'muon'][: int(rng.integers(1, 4))]
      psd = synth.make_psd(n, FS)
      tmpls = [synth.make_template(n, pre, FS, k) for k in kinds]
      fts = [build_filter(t, psd, FS, pre) for t in tmpls]
      filts = [orc.OFFilter(t, psd, FS, pre) for t in tmpls]
      x, _, _ = synth.make_traces(B, tmpls[0], psd, FS, fts[0].ampres, seed=int(rng.integers(1 << 30)), max_delay=max(1, n // 16))
      x32 = x.astype(np.float32); x64 = x32.astype(np.float64)
      lo = int(rng.integers(0, n - 2)); hi = int(rng.integers(lo + 1, n + 1))
      outside = bool(rng.integers(0, 2)); interp = bool(rng.integers(0, 2))
      fcut = float(rng.choice([5000.0, 10000.0, 20000.0]))
      engines = ['rocfft', 'auto']
      for eng in engines + ['lds']:
          try:
              plan = OFPlan(n, pre, FS, max_batch=int(rng.choice([7, 16, 64])), engine=eng)
          except Exception:
              continue
          try:
              ids = []
              for s, ft in enumerate(fts):
                  plan.set_filter(s, ft)
                  ids.append((plan.add_search(s, 'nodelay', lowchi2_fcutoff=fcut),
                              plan.add_search(s, 'delay', lowchi2_fcutoff=fcut, interpolate=interp),
                              plan.add_search(s, 'delay', lo, hi, outside, fcut)))
              w = plan.add_tdwindow(min(lo, n - 2), max(min(hi, n - 1), min(lo, n - 2) + 1))
              try:
                  out = plan.process(torch.as_tensor(x32, device='cuda')).cpu().numpy().astype(np.float64)
              except _lib.OfxError as e:
                  if 'not supported' in str(e) or 'exceeds' in str(e) or 'covers' in str(e):
                      continue
                  raise
              for s, (ft, filt) in enumerate(zip(fts, filts)):
                  tag = f'case {c} N={n} pre={pre} B={B} eng={eng}/{plan.engine} slot={s} win=[{lo},{hi}) out={outside} interp={interp}'
                  check_search(out, plan.search_offset(s, ids[s][0]), orc.process_events(filt, x64, 'nodelay', fcut), '', ft.ampres, FS, tag + ' nodelay')
                  check_search(out, plan.search_offset(s, ids[s][1]), orc.process_events(filt, x64, 'unconstrained', fcut, interpolate=interp), '', ft.ampres, FS, tag + ' delay', interpolated=interp)
                  r = orc.process_events(filt, x64, 'constrained', fcut, window_min_index=lo, window_max_index=hi, lgc_outside_window=outside)
                  if not np.any(r['index'] < 0):
                      check_search(out, plan.search_offset(s, ids[s][2]), r, '', ft.ampres, FS, tag + ' window')
          except AssertionError as e:
              bad += 1
              print('MISMATCH', str(e)[:300], flush=True)
          except Exception:
              bad += 1
              print('ERROR in case', c, n, pre, B, eng); traceback.print_exc()
          finally:
              plan.close()
      if verbose:
          print(f'case {c}: N={n} pre={pre} B={B} slots={len(kinds)} done', flush=True)
  return bad


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bad = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print('fuzz finished:', cases, 'cases,', bad, 'problems')
