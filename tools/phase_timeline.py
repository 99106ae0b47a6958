"""Per-phase timeline of k_fused from the diagnostic stamps build.
   make -C detprocess_amd/csrc variant NAME=stamps EXTRA=-DOFX_STAMPS      (-> gpurun_stamps.so)
   OFX_LIB=$PWD/gpurun_stamps.so python tools/phase_timeline.py [n_traces] [out.json]
Runs the headline plan (32768 samples, of1x1_unconstrained) twice -- two workgroups per CU
(the product configuration) and OFX_DIAG_WGPC=1 (one per CU) -- and reports, per phase of the
kernel, the mean shader cycles a workgroup spends in it, plus for the paired run how the two
workgroups of a CU overlap: the share of time both / one / none of them is in a VALU phase."""
import json, os, subprocess, sys
import numpy as np

PH = ['loop', 'wait+td', 'F1', 'E1', 'F2', 'E2', 'mid', 'E3', 'I2', 'E4', 'I1', 'tailA', 'tailB']
VALU = {'F1', 'F2', 'mid', 'I2', 'I1'}            # register-resident arithmetic phases
NT = 40
NW = 4


def run(n, wgpc, path, config3=False):
    env = dict(os.environ, OFX_STAMP_FILE=path)
    if wgpc == 1:
        env['OFX_DIAG_WGPC'] = '1'
    if config3 == 2:
        code = f'''
import sys, numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, SynthSource, search_range, utils
N=32768; fs=1.25e6; pre=N//2
tmpl=synth.make_template(N,pre,fs); psd=synth.make_psd(N,fs); ft=build_filter(tmpl,psd,fs,pre)
gen=SynthSource(N,tmpl,psd,fs,3*ft.ampres,300*ft.ampres,0.5,2000,seed=1)
x=torch.empty(({n},N),dtype=torch.float32,device='cuda:0'); gen.fill(0,{n},x)
plan=OFPlan(N,pre,fs,max_batch=8192,engine='fused'); plan.set_filter(0,ft)
lo,hi=search_range(N,pre,fs,-400,400); plan.add_search(0,'delay',lo,hi)
plan.add_tdwindow(*utils.get_window_indices(nb_samples=N,nb_pretrigger_samples=pre,fs=fs,window_min_from_trig_usec=-10,window_max_from_trig_usec=500))
plan.add_tdwindow(*utils.get_window_indices(nb_samples=N,nb_pretrigger_samples=pre,fs=fs,window_min_from_trig_usec=-500,window_max_from_trig_usec=500))
for _ in range(3): plan.process(x)
torch.cuda.synchronize()
'''
        subprocess.run([sys.executable, '-c', code], env=env, check=True)
        a = np.fromfile(path, dtype=np.uint64).reshape(-1, NT, NW, 16)
        return a[:, :, 0, :]
    if config3:
        code = f'''
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from detprocess_amd import FeatureProcessing, build_filter, synth, SynthSource
N=32768; fs=1.25e6; pre=N//2
tmpl=synth.make_template(N,pre,fs); psd=synth.make_psd(N,fs); ft=build_filter(tmpl,psd,fs,pre)
gen=SynthSource(N,tmpl,psd,fs,3*ft.ampres,300*ft.ampres,0.5,2000,seed=1)
B={n}//4
x=torch.empty((B,4,N),dtype=torch.float32,device='cuda:0'); gen.fill(0,B*4,x.reshape(B*4,N))
fp=FeatureProcessing(bench.yaml_config3(), bench.filter_data3(pre), bench.CHANNELS3, fs)
for _ in range(2): fp.process_device(x)
torch.cuda.synchronize()
'''
        subprocess.run([sys.executable, '-c', code], env=env, check=True)
        a = np.fromfile(path, dtype=np.uint64).reshape(-1, NT, NW, 16)
        return a[:, :, 0, :]
    code = f'''
import sys, numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, SynthSource
N=32768; fs=1.25e6; pre=N//2
tmpl=synth.make_template(N,pre,fs); psd=synth.make_psd(N,fs); ft=build_filter(tmpl,psd,fs,pre)
gen=SynthSource(N,tmpl,psd,fs,3*ft.ampres,300*ft.ampres,0.5,2000,seed=1)
x=torch.empty(({n},N),dtype=torch.float32,device='cuda:0'); gen.fill(0,{n},x)
plan=OFPlan(N,pre,fs,max_batch=8192,engine='fused'); plan.set_filter(0,ft); plan.add_search(0,'delay')
for _ in range(3): plan.process(x)
torch.cuda.synchronize()
'''
    subprocess.run([sys.executable, '-c', code], env=env, check=True)
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, NT, NW, 16)
    return a[:, :, 0, :]          # wave 0 of every workgroup


def summarise(a):
    t = a[:, :, :13].astype(np.int64)
    ok = (t[:, :, 12] > 0) & (t[:, :, 0] > 0)
    ok[:, :4] = False                               # warm-up traces
    d = np.diff(t, axis=2)                          # phase i = stamp i -> stamp i+1
    nxt = np.zeros_like(t[:, :, 0]); nxt[:, :-1] = t[:, 1:, 0]
    last = nxt - t[:, :, 12]                        # tail B remainder + loop overhead
    ok[:, -2:] = False
    ok &= nxt > 0
    per = {PH[i + 1]: float(d[:, :, i][ok].mean()) for i in range(12)}
    per['tailB'] = per.get('tailB', 0.0)
    per['loop'] = float(last[ok].mean())
    per['total'] = float((nxt - t[:, :, 0])[ok].mean())
    return per


def clock_mhz(a):
    """Shader clock inside the kernel: d(s_memtime) / d(s_memrealtime) x 100 MHz between the stamps 0
    of consecutive traces of a workgroup (slot 15 holds the 100 MHz counter at stamp 0)."""
    t = a[:, :, 0].astype(np.int64)
    r = a[:, :, 15].astype(np.int64)
    dt, dr = t[:, 8:-2] - t[:, 4:-6], r[:, 8:-2] - r[:, 4:-6]      # four traces apart
    ok = (dr > 0) & (dt > 0) & (t[:, 8:-2] > 0) & (t[:, 4:-6] > 0)
    if not ok.any():
        return None
    c = 100.0 * dt[ok] / dr[ok]
    return {'mean': float(c.mean()), 'p05': float(np.percentile(c, 5)), 'p95': float(np.percentile(c, 95)),
            'samples': int(ok.sum()), 'method': 'd(s_memtime) / d(s_memrealtime) x 100 MHz, four traces apart, every workgroup'}


def overlap(a):
    """Pairs of workgroups on the same CU (XCC_ID, HW_ID se/sh/cu bits): fraction of the common
    time span in which 2 / 1 / 0 of them are in a register-arithmetic phase."""
    t = a[:, :, :13].astype(np.int64)
    hw = (a[:, 4, 13].astype(np.int64)) & 0xffffffff
    xcc = a[:, 4, 14].astype(np.int64) & 0xf
    cu = (xcc << 16) | (hw & 0xff00)              # cu_id[11:8], sh_id[12], se_id[15:13]
    res = []
    for key in np.unique(cu):
        w = np.nonzero(cu == key)[0]
        if len(w) != 2:
            continue
        iv = []
        for g in w:
            segs = []
            for k in range(6, NT - 3):
                if t[g, k, 12] == 0:
                    continue
                for i, name in enumerate(PH[1:]):
                    if name in VALU:
                        segs.append((t[g, k, i], t[g, k, i + 1]))
            iv.append(segs)
        if not iv[0] or not iv[1]:
            continue
        lo = max(iv[0][0][0], iv[1][0][0]); hi = min(iv[0][-1][1], iv[1][-1][1])
        if hi <= lo:
            continue
        ev = []
        for segs in iv:
            for s, e in segs:
                s, e = max(s, lo), min(e, hi)
                if e > s:
                    ev += [(s, 1), (e, -1)]
        ev.sort()
        cnt, prev, acc = 0, lo, [0, 0, 0]
        for x, dlt in ev:
            acc[cnt] += x - prev; prev = x; cnt += dlt
        acc[cnt] += hi - prev
        res.append([v / (hi - lo) for v in acc])
    r = np.asarray(res)
    return {'pairs': int(len(r)), 'none_in_valu_phase': float(r[:, 0].mean()),
            'one_in_valu_phase': float(r[:, 1].mean()), 'both_in_valu_phase': float(r[:, 2].mean())}


def run25(n, path):
    """k_fused25 (25000 samples; make variant NAME=f25_stamps VARSRC=ofx_fused25.hip
    EXTRA='-DOFX_QUICK -DOFX_STAMPS'): stamps of every wave."""
    env = dict(os.environ, OFX_STAMP_FILE=path)
    code = f'''
import sys, numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, SynthSource
N=25000; fs=1.25e6; pre=N//2
tmpl=synth.make_template(N,pre,fs); psd=synth.make_psd(N,fs); ft=build_filter(tmpl,psd,fs,pre)
gen=SynthSource(N,tmpl,psd,fs,3*ft.ampres,300*ft.ampres,0.5,2000,seed=1)
x=torch.empty(({n},N),dtype=torch.float32,device='cuda:0'); gen.fill(0,{n},x)
plan=OFPlan(N,pre,fs,max_batch=8192,engine='fused'); plan.set_filter(0,ft); plan.add_search(0,'delay')
for _ in range(3): plan.process(x)
torch.cuda.synchronize()
'''
    subprocess.run([sys.executable, '-c', code], env=env, check=True)
    return np.fromfile(path, dtype=np.uint64).reshape(-1, NT, NW, 16)


if __name__ == '__main__':
    if '--n25000' in sys.argv:
        sys.argv.remove('--n25000')
        a = run25(int(sys.argv[1]) if len(sys.argv) > 1 else 65536, '/tmp/stamps25.bin')
        rep = {f'cycles_per_phase_wave{w}': summarise(a[:, :, w, :]) for w in range(NW)}
        json.dump(rep, open(sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/phase_timeline_25000.json', 'w'), indent=1)
        for k in PH[1:] + ['loop', 'total']:
            print(f'{k:8s} ' + '  '.join(f"w{w} {rep[f'cycles_per_phase_wave{w}'][k]:8.0f}" for w in range(NW)))
        # sub-stamps of the tail: 10 -> 13 (max, first reduction) -> 14 (arg-max) -> 15 (windows, bands) -> 11 (lowchi2)
        tt = a[:, 6:NT - 3, :, :].astype(np.int64)
        ok = (tt[..., 12] > 0) & (tt[..., 13] > 0)
        seq = [10, 13, 14, 15, 11]
        for i in range(4):
            dd = (tt[..., seq[i + 1]] - tt[..., seq[i]])
            print(f'tail {seq[i]:2d}->{seq[i + 1]:2d} ' + '  '.join(f'w{w} {dd[:, :, w][ok[:, :, w]].mean():8.0f}' for w in range(NW)))
        sys.exit(0)
    if '--config2' in sys.argv:
        sys.argv.remove('--config2')
        a2 = run(int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 2, '/tmp/stamps2c.bin', config3=2)
        rep = {'workload': 'config2', 'cycles_per_phase_config2': summarise(a2), 'overlap_2wg': overlap(a2),
               'clock_mhz': clock_mhz(a2)}
        json.dump(rep, open(sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/phase_timeline_c2.json', 'w'), indent=1)
        for k, v in rep['cycles_per_phase_config2'].items():
            print(f'{k:8s} {v:9.0f}')
        print(rep['overlap_2wg'])
        sys.exit(0)
    if '--config3' in sys.argv:
        # BASELINE configs[3] (three slots, nine searches, windows, bands): the stamps of a trace
        # cover its last slot pass only (markers 5..12 are re-stamped per slot), the total is exact
        sys.argv.remove('--config3')
        a3 = run(int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 2, '/tmp/stamps3.bin', config3=True)
        rep = {'workload': 'config3', 'cycles_per_phase_config3_last_slot': summarise(a3), 'clock_mhz': clock_mhz(a3)}
        json.dump(rep, open(sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/phase_timeline_c3.json', 'w'), indent=1)
        for k, v in rep['cycles_per_phase_config3_last_slot'].items():
            print(f'{k:8s} {v:9.0f}')
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    dest = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/phase_timeline.json'
    a2 = run(n, 2, '/tmp/stamps2.bin')
    a1 = run(n, 1, '/tmp/stamps1.bin')
    rep = {'workload': 'config1', 'cycles_per_phase_2wg_per_cu': summarise(a2), 'cycles_per_phase_1wg_per_cu': summarise(a1),
           'overlap_2wg': overlap(a2), 'clock_mhz': clock_mhz(a2), 'clock_mhz_1wg_per_cu': clock_mhz(a1)}
    json.dump(rep, open(dest, 'w'), indent=1)
    for k in PH[1:] + ['loop', 'total']:
        print(f"{k:8s} 2wg {rep['cycles_per_phase_2wg_per_cu'][k]:9.0f}   1wg {rep['cycles_per_phase_1wg_per_cu'][k]:9.0f}")
    print(rep['overlap_2wg'])
    print('clock in the kernel (MHz):', rep.get('clock_mhz'))
