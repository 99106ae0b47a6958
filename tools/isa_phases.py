"""Instruction mix per phase of the FUSED kernel, from the ";ofxphase i" markers in its ISA.
   hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -Iinclude -S --cuda-device-only \
         detprocess_amd/csrc/ofx_fused.hip -o /tmp/fused.s && python tools/isa_phases.py /tmp/fused.s
(the scheduler moves instructions across the markers, so the boundaries are approximate)."""
import re,sys
f = sys.argv[1]
src = open(f).read().split('\n')
feat = sys.argv[2] if len(sys.argv) > 2 else '0'
st = next(i for i,l in enumerate(src) if l.startswith((sys.argv[3] if len(sys.argv) > 3 else '_ZN12_GLOBAL__N_17k_fusedILi') + feat + 'ELb0E'))
end = next(i for i,l in enumerate(src) if '.amdhsa_kernel' in l and i>st)
lines = src[st:end]
def new(): return {'valu':0,'pk':0,'lds':0,'vmem':0,'wait':0,'vm0':0,'salu':0,'bar':0,'scr':0,'smem':0,'rdln':0,'mov':0}
seg=[]; cur=new(); names=['pre']
for l in lines:
    t=l.strip()
    m = re.match(r';ofxphase (\d+)', t)
    if m: seg.append(cur); cur=new(); names.append('after'+m.group(1)); continue
    if t.startswith('v_readlane') or t.startswith('v_writelane'): cur['rdln']+=1
    elif t.startswith('v_pk_'): cur['pk']+=1; cur['valu']+=1
    elif t.startswith('v_mov'): cur['mov']+=1; cur['valu']+=1
    elif t.startswith('v_'): cur['valu']+=1
    elif t.startswith('ds_'): cur['lds']+=1
    elif t.startswith('buffer_') or t.startswith('global_'): cur['vmem']+=1
    elif t.startswith('scratch_'): cur['scr']+=1
    elif t.startswith('s_waitcnt'):
        cur['wait']+=1
        if 'vmcnt(0)' in t: cur['vm0']+=1
    elif t.startswith('s_barrier'): cur['bar']+=1
    elif t.startswith('s_load') or t.startswith('s_buffer_load'): cur['smem']+=1
    elif t.startswith('s_'): cur['salu']+=1
seg.append(cur)
nm={'pre':'pre','after0':'load+td','after1':'F1','after2':'E1','after3':'F2','after4':'E2','after5':'mid','after6':'E3','after7':'I2','after8':'E4','after9':'I1','after10':'tailA','after11':'tailB','after12':'looptail'}
for n,s in zip(names,seg): print(f"{nm.get(n,n):9s}", ' '.join(f"{k}={v}" for k,v in s.items() if v))
