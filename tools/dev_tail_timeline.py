"""Sub-phases of the tail of k_fused on BASELINE configs[3] (last slot of the slot loop), from the
stamps build with the tail stamps:
   make -C detprocess_amd/csrc variant NAME=tstamps EXTRA='-DOFX_STAMPS -DOFX_TAILSTAMPS'
   OFX_LIB=$PWD/gpurun_tstamps.so python tools/dev_tail_timeline.py [n_traces] [out.json]"""
import json, os, subprocess, sys
import numpy as np
NT, NW = 40, 4
ORDER = [(0, 'loop start'), (1, 'trace wait + windows'), (10, 'transforms + earlier slots'),
         (2, 'group maxima, table requests'), (3, 'block max + chi2_0 (2 barriers)'),
         (4, 'full-range arg-max'), (5, 'lag dump + window scans'), (6, 'bands / interpolation'),
         (7, 'lowchi2: bins in LDS'), (11, 'lowchi2: stash bins (WIDE)'), (8, 'barrier + row write'),
         (12, 'next request issued')]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
out = sys.argv[2] if len(sys.argv) > 2 else None
path = '/tmp/ofx_tstamps.bin'
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
subprocess.run([sys.executable, '-c', code], env=dict(os.environ, OFX_STAMP_FILE=path), check=True)
a = np.fromfile(path, dtype=np.uint64).reshape(-1, NT, NW, 16)[:, :, 0, :].astype(np.int64)
ok = (a[:, :, 12] > 0) & (a[:, :, 0] > 0)
ok[:, :4] = False
ok[:, -2:] = False
res = {}
for (i0, _), (i1, name) in zip(ORDER[:-1], ORDER[1:]):
    d = (a[:, :, i1] - a[:, :, i0])[ok]
    res[name] = float(d.mean())
nxt = np.zeros_like(a[:, :, 0]); nxt[:, :-1] = a[:, 1:, 0]
okn = ok & (nxt > 0)
res['total per event'] = float((nxt - a[:, :, 0])[okn].mean())
for k, v in res.items():
    print(f'{k:36s} {v:10.0f} cycles')
if out:
    json.dump(res, open(out, 'w'), indent=1)
