"""Per-phase timeline of k_wave (4096-sample traces, one wave per trace) from the diagnostic stamps build.
   make -C detprocess_amd/csrc variant NAME=wstamps EXTRA=-DOFX_STAMPS VARSRC=ofx_wave.hip
   OFX_LIB=$PWD/gpurun_wstamps.so python tools/wave_timeline.py [n_traces] [out.json] [workload]
workload: 1 = of1x1_unconstrained (default), 2 = constrained + two windows.
Every stamp of this build waits for all outstanding memory operations first (s_waitcnt 0), so a phase
carries the latencies it started; the product build overlaps more than this timeline shows.  Reports
the mean shader cycles of a wave in every phase over its traces 3 .. 30."""
import json, os, subprocess, sys
import numpy as np

PH = ['load', 'F1', 'E1', 'F2', 'E2', 'F3', 'mid', 'I3', 'E3+I2', 'E4', 'I1+dump', 'bands', 'searches+row', 'loop']
NT = 40


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
    out = sys.argv[2] if len(sys.argv) > 2 else None
    wl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    path = '/tmp/wave_stamps.bin'
    extra = ''
    if wl == 2:
        extra = '''
lo,hi=search_range(N,pre,fs,-400,400); plan.add_search(0,'delay',lo,hi)
plan.add_tdwindow(100, 1500); plan.add_tdwindow(1800, 2700)
'''
    else:
        extra = "plan.add_search(0,'delay')\n"
    code = f'''
import sys, numpy as np, torch
sys.path.insert(0, '.')
from detprocess_amd import OFPlan, build_filter, synth, SynthSource, search_range
N=4096; fs=1.25e6; pre=N//2
tmpl=synth.make_template(N,pre,fs); psd=synth.make_psd(N,fs); ft=build_filter(tmpl,psd,fs,pre)
gen=SynthSource(N,tmpl,psd,fs,3*ft.ampres,300*ft.ampres,0.5,200,seed=1)
x=torch.empty(({n},N),dtype=torch.float32,device='cuda:0'); gen.fill(0,{n},x)
plan=OFPlan(N,pre,fs,max_batch={n},engine='fused'); plan.set_filter(0,ft)
{extra}
for _ in range(3): plan.process(x)
torch.cuda.synchronize()
'''
    subprocess.run([sys.executable, '-c', code], env=dict(os.environ, OFX_STAMP_FILE=path), check=True)
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, NT, 16).astype(np.int64)
    t = a[:, 3:31, :13]                       # [wave, trace, stamp]
    ok = (t > 0).all(axis=(1, 2))
    t = t[ok]
    d = np.diff(t, axis=2)                    # 12 phases within a trace
    loop = t[:, 1:, 0] - t[:, :-1, 12]        # end of a trace to the top of the next
    rows = {PH[i]: float(d[:, :, i].mean()) for i in range(12)}
    rows[PH[13]] = float(loop.mean())
    total = float((t[:, 1:, 0] - t[:, :-1, 0]).mean())
    rt = a[ok][:, 3:31, 15]
    clk = float(((t[:, -1, 0] - t[:, 0, 0]) / np.maximum(rt[:, -1] - rt[:, 0], 1)).mean() * 100.0)
    rec = {'kernel': 'k_wave', 'workload': f'config{wl}_n4096', 'n_traces': n,
           'waves': int(ok.sum()), 'cycles_per_trace_per_wave': total,
           'clock_mhz': {'mean': clk, 'how': 'd s_memtime / d s_memrealtime x 100 MHz between the stamps 0 of '
                                               'the traces 3 and 30 of every wave'},
           'phases': rows,
           'note': 'stamps wait for outstanding memory operations (s_waitcnt 0) -- latencies are exposed'}
    print(json.dumps(rec, indent=1))
    if out:
        json.dump(rec, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main()
