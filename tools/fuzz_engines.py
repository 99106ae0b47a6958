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
              wlo, whi = min(lo, n - 2), max(min(hi, n - 1), min(lo, n - 2) + 1)
              w = plan.add_tdwindow(wlo, whi)
              klo = int(rng.integers(1, max(2, min(200, n // 4))))
              khi = int(rng.integers(klo + 1, min(n // 2, klo + 300) + 1))
              band = plan.add_band(klo, khi)
              try:
                  out = plan.process(torch.as_tensor(x32, device='cuda')).cpu().numpy().astype(np.float64)
              except _lib.OfxError as e:
                  if 'not supported' in str(e) or 'exceeds' in str(e) or 'covers' in str(e):
                      continue
                  raise
              # time-domain window and psd_amp band of the (single-channel) events
              tagw = f'case {c} N={n} pre={pre} B={B} eng={eng}/{plan.engine} td=[{wlo},{whi}) band=[{klo},{khi})'
              seg = x64[:, wlo:whi]; sc = np.abs(x64).max(); ow = plan.tdwindow_offset(w)
              assert np.allclose(out[:, ow + 0], seg.mean(axis=1), rtol=1e-4, atol=3e-6 * sc), tagw + ' baseline'
              assert np.allclose(out[:, ow + 1], (seg.sum(axis=1) - 0.5 * (seg[:, 0] + seg[:, -1])) / FS, rtol=1e-4,
                                 atol=3e-6 * sc * (whi - wlo) / FS), tagw + ' integral'
              assert np.array_equal(out[:, ow + 2], seg.max(axis=1)) and np.array_equal(out[:, ow + 3], seg.min(axis=1)), \
                  tagw + ' max/min'
              V = np.fft.rfft(x64, axis=-1) / n
              fold = np.abs(V) ** 2 * n / FS
              fold[:, 1:(None if n % 2 else -1)] *= 2.0
              want_band = np.sqrt(fold[:, klo:khi]).mean(axis=1)
              assert np.allclose(out[:, plan.band_offset(band)], want_band, rtol=2e-4), tagw + ' psd_amp band'
              for s, (ft, filt) in enumerate(zip(fts, filts)):
                  tag = f'case {c} N={n} pre={pre} B={B} eng={eng}/{plan.engine} slot={s} win=[{lo},{hi}) out={outside} interp={interp}'
                  check_search(out, plan.search_offset(s, ids[s][0]), orc.process_events(filt, x64, 'nodelay', fcut), '', ft.ampres, FS, tag + ' nodelay')
                  r1 = orc.process_events(filt, x64, 'unconstrained', fcut, interpolate=interp)
                  # interpolated fits are checked on clear pulses only: the fp32 error of the sub-sample offset
                  # scales as 1 / SNR (1e-3 sample at SNR ~ 30, the tolerance of tests/util.py)
                  keep = np.abs(r1['amp']) > 50 * ft.ampres if interp else np.ones(B, bool)
                  if keep.any():
                      o1 = plan.search_offset(s, ids[s][1])
                      dts = np.abs(out[keep, o1 + 1] - r1['t0'][keep]) * FS
                      wz = int(np.argmax(dts))
                      check_search(out[keep], o1, {kk: np.asarray(v)[keep] for kk, v in r1.items()}, '', ft.ampres, FS,
                                   tag + f' delay (worst t0 error {dts[wz]:.2e} samples at snr {abs(r1["amp"][keep][wz]) / ft.ampres:.1f}, '
                                         f'idx {out[keep, o1 + 7][wz]:.0f}/{r1["index"][keep][wz]}, t0 {out[keep, o1 + 1][wz]:.6e}/{r1["t0"][keep][wz]:.6e})',
                                   interpolated=interp)
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


def run_nxm(cases, seed, verbose=True):
    """N x M engine: random lengths (rocFFT and every LDS-transform build), channel / template
    counts, channel maps, valid masks, windows."""
    from detprocess_amd.ofnxm import NxMPlan, build_nxm_filter
    from oracle import ofnxm as onm
    import test_ofnxm as T
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        n = int(rng.choice([2 * int(rng.integers(64, 1500)), int(rng.choice([250, 1000, 1024, 2000, 4096, 5000, 6250, 8192,
                                                                             12000, 16384, 20000, 25000, 30000]))]))
        pre = int(rng.integers(n // 8, n - n // 8))
        C, M = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        B = int(rng.integers(1, 40)) if n <= 8192 else int(rng.integers(1, 7))
        tm = T.make_templates(n, pre, C, M)
        csd = T.make_csd(n, C)
        filt = onm.NxMFilter(tm, csd, FS, pre)
        if np.linalg.cond(filt.P) > 50.0:      # near-degenerate template sets amplify fp32 rounding
            continue                           # beyond the stated tolerance (not an engine property)
        ev, _, _ = T.make_events(B, tm, csd, filt.ampres, seed=int(rng.integers(1 << 30)), max_delay=max(1, n // 16))
        n_total = int(rng.integers(C, C + 3))
        idx = rng.permutation(n_total)[:C]
        full = rng.normal(0, 1e-7, (B, n_total, n)).astype(np.float32)
        full[:, idx] = ev.astype(np.float32)
        valid = (rng.random(B) < 0.85).astype(np.uint8)
        lo = int(rng.integers(0, n - 2)); hi = int(rng.integers(lo + 1, n + 1))
        outside = bool(rng.integers(0, 2))
        tag = f'nxm case {c} N={n} pre={pre} {C}x{M} B={B} map={list(idx)}/{n_total} win=[{lo},{hi}) out={outside}'
        plan = NxMPlan(build_nxm_filter(tm, csd, FS, pre), max_batch=int(rng.choice([3, 8, 64])))
        try:
            plan.set_channels(n_total, idx)
            s0 = plan.add_search('nodelay'); s1 = plan.add_search('delay'); s2 = plan.add_search('delay', lo, hi, outside)
            out = plan.process(torch.as_tensor(full, device='cuda'), torch.as_tensor(valid, device='cuda')).cpu().numpy()
            ok = valid.astype(bool)
            assert np.all(out[~ok] == -999999.0), tag + ' sentinel rows'
            if ok.any():
                x = full[ok][:, idx].astype(np.float64)
                r = onm.process_events(filt, x)
                T._check(plan, out[ok], s0, r, filt, nodelay=True)
                T._check(plan, out[ok], s1, r, filt)
                rw = onm.process_events(filt, x, window_min_index=lo, window_max_index=hi, lgc_outside_window=outside)
                if np.all(rw['index'] >= 0):
                    T._check(plan, out[ok], s2, rw, filt)
        except AssertionError as e:
            bad += 1
            print('MISMATCH', tag, str(e)[:200], flush=True)
        except Exception:
            bad += 1
            print('ERROR', tag); traceback.print_exc()
        finally:
            plan.close()
        if verbose:
            print(tag, 'done', flush=True)
    return bad


def run_trigger(cases, seed, verbose=True):
    """Trigger stage: random filter lengths, stream lengths around the
    overlap-save block boundaries, channel / amplitude counts, padding on and off."""
    from detprocess_amd import OptimumFilterTrigger
    from oracle import oftrigger as ot
    import test_ofnxm as T
    rng = np.random.default_rng(seed)
    bad = 0
    extended = 0
    for c in range(cases):
        nxm = bool(rng.integers(0, 2))
        n = 2 * int(rng.integers(32, 1500))              # even: odd trace lengths are rejected
        pre = int(rng.integers(n // 8, n - n // 8))
        P = 4096
        while P < 4 * n:
            P *= 2
        H = P - (n - 1)
        kblk = int(rng.integers(1, 6))
        L = int(max(n + 2, kblk * H + int(rng.integers(-3, 4)) - int(rng.choice([0, (n - 1) // 2, n - 1]))))
        padding = bool(rng.integers(0, 2))
        C, M = (int(rng.integers(1, 4)), int(rng.integers(1, 4))) if nxm else (1, 1)
        tag = f'trigger case {c} N={n} pre={pre} L={L} {C}x{M} padding={padding}'
        try:
            if nxm:
                tm = T.make_templates(n, pre, C, M)
                csd = T.make_csd(n, C)
                tr = ot.OFTriggerNxM(FS, tm, csd, pre)
                if np.linalg.cond(tr.w_matrix) > 50.0:
                    continue
                x = np.zeros((C, L))
                for a in range(C):
                    x[a] = synth.coloured_noise(rng, L // n + 2, csd[a, a].real, FS).reshape(-1)[:L]
                for p0 in rng.integers(0, max(1, L - n), 3):
                    x[:, p0:p0 + n] += np.einsum('m,amn->an', tr.resolution * rng.uniform(5, 50, M), tm)[:, :L - p0]
                g = OptimumFilterTrigger(list('abc'[:C]), FS, tm, csd, pre)
            else:
                tmpl = synth.make_template(n, pre, FS)
                psd = synth.make_psd(n, FS)
                tr = ot.OFTrigger(FS, tmpl, psd, pre)
                x = synth.coloured_noise(rng, L // n + 2, psd, FS).reshape(-1)[:L]
                for p0 in rng.integers(0, max(1, L - n), 3):
                    x[p0:p0 + n] += (tr.resolution * rng.uniform(5, 50) * tmpl)[:L - p0]
                g = OptimumFilterTrigger('a', FS, tmpl, psd, pre)
            x32 = x.astype(np.float32)
            filt, dchi = tr.update_trace(x32.astype(np.float64), padding=padding)
            g.update_trace(x32, padding=padding)
            gf = g.get_filtered_trace().astype(np.float64)
            gd = g.get_filtered_delta_chi2().astype(np.float64)
            filt = np.atleast_2d(filt)
            scale = np.max(np.abs(filt), axis=1, keepdims=True) + 1e-300
            assert gf.shape == filt.shape, tag + f' shape {gf.shape} vs {filt.shape}'
            assert np.all(np.abs(gf - filt) <= 4e-5 * scale), tag + f' filtered {np.max(np.abs(gf - filt) / scale):.2e}'
            assert np.max(np.abs(gd - dchi)) <= 8e-5 * max(np.max(dchi), 1e-300), tag + ' delta chi2'
            # dynamic pile-up window and residual pass (padded traces: every trigger is at least
            # a template length from the edges, where the reference's slices are well defined)
            dmax = float(np.max(dchi))
            if padding and dmax > 0 and (not nxm or C == M) and L > 3 * n:
                thr_s = float(rng.choice([4.0, 6.0]))
                w0, w1 = float(rng.integers(0, 40)), float(rng.integers(0, n))
                fn = lambda d: w0 + w1 * min(1.0, d / dmax)
                ref = tr.find_triggers(thr_s, dynamic_function=fn)
                g.find_triggers(thr_s, dynamic=True, dynamic_threshold_function=fn)
                got = g.get_trigger_data()[g._trigger_name]['trigger_index']
                thr = ref['chi2_threshold']
                clear = {int(i) for i, d in zip(ref['trigger_index'], ref['trigger_delta_chi2']) if d > 2 * thr + 1e-3 * dmax}
                assert clear <= set(got), tag + f' dynamic: missing {sorted(clear - set(got))[:5]}'
                assert abs(len(got) - len(ref['trigger_index'])) <= max(3, len(got) // 5), tag + ' dynamic: count'
                win = int(rng.integers(0, n))
                first, second, residual, combined = tr.find_triggers_residual(thr_s, x32.astype(np.float64),
                                                                              pileup_window_samples=win)
                out = g.find_triggers(thr_s, pileup_window_samples=win, residual=True, return_trigger_data=True)
                assert np.max(np.abs(out[3].astype(np.float64) - residual)) <= 2e-4 * dmax, \
                    tag + f' residual trace {np.max(np.abs(out[3] - residual)) / dmax:.2e}'
                assert np.array_equal(g.get_filtered_delta_chi2(), gd.astype(np.float32)), tag + ' trace not restored'
                c2 = {int(i) for i, d in zip(second['trigger_index'], second['trigger_delta_chi2']) if d > 2 * thr + 1e-3 * dmax}
                assert c2 <= set(out[2][g._trigger_name]['trigger_index']), tag + ' residual: second pass'
                extended += 1
            g.close()
        except AssertionError as e:
            bad += 1
            print('MISMATCH', str(e)[:200], flush=True)
        except Exception:
            bad += 1
            print('ERROR', tag); traceback.print_exc()
        if verbose:
            print(tag, 'done', flush=True)
    print(f'trigger: dynamic window + residual pass checked in {extended} cases', flush=True)
    return bad


def run_fused(cases, seed, verbose=True, n=32768):
    """The FUSED kernel at 32768 samples (mode 'fused') or 25000 samples (mode 'fused25', k_fused25) under random plans (1-3 template tags, no-delay /
    full / windowed / outside / interpolated fits, 0-3 time-domain windows, channel sums with
    weights, valid masks, batches below and above the persistent grid) against the ROCFFT
    engine on every event and against the oracle on a few."""
    rng = np.random.default_rng(seed)
    bad = 0
    psd = synth.make_psd(n, FS)
    for c in range(cases):
        pre = int(rng.choice([n // 2, n // 2, int(rng.integers(n // 8, n - n // 8))]))
        kinds = ['pulse', 'glitch', 'muon'][: int(rng.integers(1, 4))]
        tmpls = [synth.make_template(n, pre, FS, k) for k in kinds]
        fts = [build_filter(t, psd, FS, pre) for t in tmpls]
        filts = [orc.OFFilter(t, psd, FS, pre) for t in tmpls]
        B = int(rng.choice([3, 70, 600, 1100]))
        n_total = int(rng.integers(1, 4))
        nterm = int(rng.integers(1, min(2, n_total) + 1))
        chans = list(rng.permutation(n_total)[:nterm])
        weights = [1.0] if (nterm == 1 and rng.integers(0, 2)) else list(rng.choice([1.0, -1.0, 0.7, 1.3], nterm))
        x, _, _ = synth.make_traces(B * n_total, tmpls[0], psd, FS, fts[0].ampres, seed=int(rng.integers(1 << 30)),
                                    max_delay=min(2000, n // 4))
        ev = x.reshape(B, n_total, n).astype(np.float32)
        valid = (rng.random(B) < 0.9).astype(np.uint8)
        searches = []
        for s in range(len(kinds)):
            ss = []
            if rng.integers(0, 2): ss.append(('nodelay', 0, n, False, False))
            if rng.integers(0, 2): ss.append(('delay', 0, n, False, bool(rng.integers(0, 2))))
            if rng.integers(0, 2) or not ss:
                lo = int(rng.integers(0, n - 2)); hi = int(rng.integers(lo + 1, min(n, lo + int(rng.choice([50, 1000, 20000]))) + 1))
                ss.append(('delay', lo, hi, bool(rng.integers(0, 4) == 0), bool(rng.integers(0, 2))))
            searches.append(ss)
        tdw = []
        for _ in range(int(rng.integers(0, 4))):
            lo = int(rng.integers(0, n - 2)); tdw.append((lo, int(rng.integers(lo + 1, n))))
        # beyond 512 bins: the stash (32768 samples); 25000 samples: up to the 1250 bins kept in LDS
        fcut_c = float(rng.choice([10000.0, 10000.0, 50000.0, {32768: 120000.0, 4096: 70000.0, 8192: 70000.0, 16384: 70000.0}.get(n, 62000.0)]))
        tag = f'fused case {c} pre={pre} slots={len(kinds)} B={B} chans={chans}/{n_total} w={weights} td={len(tdw)} fcut={fcut_c}'
        if verbose:
            print('   searches', searches, 'windows', tdw, flush=True)
        outs = {}
        try:
            for eng in ('fused', 'rocfft'):
                plan = OFPlan(n, pre, FS, max_batch=int(rng.choice([64, 8192])), engine=eng)
                ids = []
                for s, ft in enumerate(fts):
                    plan.set_filter(s, ft)
                    ids.append([plan.add_search(s, k, lo, hi, outside, fcut_c, interp) for (k, lo, hi, outside, interp) in searches[s]])
                wids = [plan.add_tdwindow(lo, hi) for lo, hi in tdw]
                if n_total > 1 or nterm > 1 or weights[0] != 1.0:
                    plan.set_channels(n_total, chans, weights)
                inp = torch.as_tensor(ev if n_total > 1 or nterm > 1 or weights[0] != 1.0 else ev[:, 0], device='cuda')
                outs[eng] = plan.process(inp, valid=torch.as_tensor(valid, device='cuda')).cpu().numpy().astype(np.float64)
                offs = [[plan.search_offset(s, q) for q in ids[s]] for s in range(len(kinds))]
                woffs = [plan.tdwindow_offset(w) for w in wids]
                plan.close()
            a, b = outs['fused'], outs['rocfft']
            ok = valid.astype(bool)
            assert np.all(a[~ok] == -999999.0) and np.all(b[~ok] == -999999.0), tag + ' sentinel rows'
            comb = np.zeros((B, n))
            for ch, w in zip(chans, weights):
                comb += np.float32(w).astype(np.float64) * ev[:, ch].astype(np.float64)
            for s in range(len(kinds)):
                for q, (k, lo, hi, outside, interp) in enumerate(searches[s]):
                    o = offs[s][q]
                    flips = a[ok, o + 7] != b[ok, o + 7]
                    # a flipped bin must be a near tie: both engines see the same chi2 there
                    if flips.sum() > max(1, 0.01 * flips.size):
                        ev_ = np.nonzero(ok)[0][flips][:10]
                        print('   flipped events', [(int(e_), int(a[e_, o + 7]), int(b[e_, o + 7]), float(a[e_, o + 2]), float(b[e_, o + 2]),
                                                      float(a[e_, o]), float(b[e_, o])) for e_ in ev_], flush=True)
                        ro_ = orc.process_events(filts[s], comb[ev_[:4]], 'nodelay' if k == 'nodelay' else 'constrained', fcut_c,
                                                 **({} if k == 'nodelay' else dict(window_min_index=lo, window_max_index=hi, lgc_outside_window=outside)))
                        print('   oracle there: idx', ro_['index'], 'chi2', ro_['chi2'], 'amp', ro_['amp'], flush=True)
                    assert flips.sum() <= max(1, 0.01 * flips.size), tag + f' slot {s} search {q}: {flips.sum()} bin flips'
                    if flips.any():
                        # (an interpolated fit reports the chi2 of the parabola's vertex: two discrete bins that tie to
                        # 1e-6 give vertices whose chi2 differ by the fp32 error of the neighbouring amplitudes)
                        assert np.allclose(a[ok][flips, o + 2], b[ok][flips, o + 2], rtol=1e-4 if interp else 1e-5), \
                            tag + f' flip is not a tie s{s} q{q}'
                    same = ~flips
                    da = np.abs(a[ok][same, o] - b[ok][same, o])
                    lim = 1e-4 * np.abs(b[ok][same, o]) + 2e-4 * fts[s].ampres
                    if not np.all(da <= lim):
                        w_ = int(np.argmax(da / lim))
                        e_ = int(np.nonzero(ok)[0][same][w_])
                        r_ = orc.process_events(filts[s], comb[e_:e_ + 1], 'nodelay' if k == 'nodelay' else 'constrained', fcut_c,
                                                interpolate=interp, **({} if k == 'nodelay' else dict(
                                                    window_min_index=lo, window_max_index=hi, lgc_outside_window=outside)))
                        r0_ = orc.process_events(filts[s], comb[e_:e_ + 1], 'nodelay' if k == 'nodelay' else 'constrained', fcut_c,
                                                 interpolate=False, **({} if k == 'nodelay' else dict(
                                                     window_min_index=lo, window_max_index=hi, lgc_outside_window=outside)))
                        print(f'   oracle amp {r_["amp"][0]:.4e} t0 {r_["t0"][0]:.6e} idx {r_["index"][0]} | no-interp amp {r0_["amp"][0]:.4e} | '
                              f'fused t0 {a[e_, o + 1]:.6e} idx {a[e_, o + 7]} rocfft t0 {b[e_, o + 1]:.6e} idx {b[e_, o + 7]}', flush=True)
                        raise AssertionError(tag + f' amp s{s} q{q} {searches[s][q]}: worst {np.max(da / lim):.1f}x at event '
                                             f'{np.nonzero(ok)[0][same][w_]} a={a[ok][same, o][w_]:.4e} b={b[ok][same, o][w_]:.4e} '
                                             f'sigma={fts[s].ampres:.3e}')
                    assert np.allclose(a[ok][:, o + 2], b[ok][:, o + 2], rtol=2e-4), tag + f' chi2 s{s} q{q}'
                    sel = np.nonzero(ok)[0][:4]
                    mode = 'nodelay' if k == 'nodelay' else 'constrained'
                    r = orc.process_events(filts[s], comb[sel], mode, fcut_c, interpolate=interp,
                                           **({} if k == 'nodelay' else dict(window_min_index=lo, window_max_index=hi, lgc_outside_window=outside)))
                    if np.all(r['index'] >= 0):
                        # the sub-sample offset of a noise peak is a ratio of amplitude differences far
                        # below fp32 resolution: interpolated fits are checked on clear pulses only
                        keep = np.abs(r['amp']) > 50 * fts[s].ampres if interp else np.ones(len(sel), bool)
                        if keep.any():
                            rk = {kk: np.asarray(v)[keep] for kk, v in r.items()}
                            check_search(a[sel][keep], o, rk, '', fts[s].ampres, FS,
                                         tag + f' oracle s{s} q{q} {searches[s][q]}', interpolated=interp)
            for w, (lo, hi) in zip(woffs, tdw):
                seg = comb[ok][:, lo:hi]
                sc = np.abs(comb).max()
                assert np.allclose(a[ok][:, w + 0], seg.mean(axis=1), rtol=1e-4, atol=3e-6 * sc), tag + ' baseline'
                assert np.allclose(a[ok][:, w + 2], seg.max(axis=1), rtol=1e-6, atol=1e-6 * sc), tag + ' maximum'
                assert np.allclose(a[ok][:, w + 3], seg.min(axis=1), rtol=1e-6, atol=1e-6 * sc), tag + ' minimum'
                assert np.allclose(a[ok][:, w:w + 4], b[ok][:, w:w + 4], rtol=1e-4, atol=3e-6 * sc), tag + ' td vs rocfft'
        except AssertionError as e:
            bad += 1
            print('MISMATCH', str(e)[:300], flush=True)
        except Exception:
            bad += 1
            print('ERROR', tag); traceback.print_exc()
        if verbose:
            print(tag, 'done', flush=True)
    return bad


def run_adc(cases, seed, verbose=True):
    """Raw-data front end: events cut on the GPU out of int16 streams (windows hanging over
    either end included) equal the same windows cut and converted on the host, bit for bit."""
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        n = 2 * int(rng.integers(64, 3000))
        pre = int(rng.integers(1, n - 1))
        C = int(rng.integers(1, 5))
        n_stream = int(rng.integers(n, 12 * n))
        B = int(rng.integers(1, 60))
        tag = f'adc case {c} N={n} pre={pre} C={C} stream={n_stream} B={B}'
        adc = rng.integers(-32768, 32768, (C, n_stream), dtype=np.int16)
        trig = rng.integers(-n, n_stream + n, B).astype(np.int64)
        trig[: min(B, 4)] = [pre, n_stream - (n - pre), pre - 1, n_stream - (n - pre) + 1][: min(B, 4)]
        scale = rng.uniform(1e-12, 5e-12, C)
        offset = rng.uniform(-1e-9, 1e-9, C)
        tmpl = synth.make_template(n, pre, FS)
        psd = synth.make_psd(n, FS)
        plan = OFPlan(n, pre, FS, max_batch=int(rng.choice([5, 32])), engine=str(rng.choice(['auto', 'rocfft'])))
        try:
            plan.set_filter(0, build_filter(tmpl, psd, FS, pre))
            plan.add_search(0, 'delay')
            plan.add_tdwindow(0, n - 1)
            ch = int(rng.integers(0, C))
            plan.set_channels(C, [ch], [1.0])
            got = plan.process_adc(adc, trig, scale, offset)
            lo = trig - pre
            ok = (lo >= 0) & (lo + n <= n_stream)
            ev = np.zeros((B, C, n), dtype=np.float32)
            for b in np.nonzero(ok)[0]:
                for k in range(C):
                    ev[b, k] = adc[k, lo[b]:lo[b] + n].astype(np.float32) * np.float32(scale[k]) + np.float32(offset[k])
            want = plan.process(ev, valid=ok.astype(np.uint8))
            assert np.array_equal(got, want), tag + f' differ in rows {np.nonzero(np.any(got != want, axis=1))[0][:5]}'
            assert np.all(got[~ok] == -999999.0), tag + ' sentinel'
            dev = plan.process_adc(torch.as_tensor(adc, device='cuda'), trig, scale, offset)
            assert np.array_equal(dev.cpu().numpy(), want), tag + ' device-resident stream'
        except AssertionError as e:
            bad += 1
            print('MISMATCH', str(e)[:200], flush=True)
        except Exception:
            bad += 1
            print('ERROR', tag); traceback.print_exc()
        finally:
            plan.close()
        if verbose:
            print(tag, 'done', flush=True)
    return bad


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3 and sys.argv[3] == 'nxm':
        bad = run_nxm(cases, seed)
    elif len(sys.argv) > 3 and sys.argv[3] == 'trigger':
        bad = run_trigger(cases, seed)
    elif len(sys.argv) > 3 and sys.argv[3] == 'fused':
        bad = run_fused(cases, seed)
    elif len(sys.argv) > 3 and sys.argv[3] == 'fused25':
        bad = run_fused(cases, seed, n=25000)
    elif len(sys.argv) > 3 and sys.argv[3] == 'fused12':
        bad = run_fused(cases, seed, n=12500)
    elif len(sys.argv) > 3 and sys.argv[3] == 'fused20':
        bad = run_fused(cases, seed, n=20000)
    elif len(sys.argv) > 3 and sys.argv[3] == 'wave':          # k_wave, 4096 samples
        bad = run_fused(cases, seed, n=4096)
    elif len(sys.argv) > 3 and sys.argv[3] == 'wave2':         # k_wave2, 8192 samples
        bad = run_fused(cases, seed, n=8192)
    elif len(sys.argv) > 3 and sys.argv[3] == 'wave4':         # k_wave2 with four waves, 16384 samples
        bad = run_fused(cases, seed, n=16384)
    elif len(sys.argv) > 3 and sys.argv[3] == 'adc':
        bad = run_adc(cases, seed)
    else:
        bad = run(cases, seed)
    print('fuzz finished:', cases, 'cases,', bad, 'problems')
