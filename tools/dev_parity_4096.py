import sys, json, numpy as np, torch
sys.path.insert(0,'.')
from detprocess_amd import OFPlan, build_filter, synth
from oracle import of1x1 as orc
FS=1.25e6; n=4096; pre=n//2; b=8192
tmpl=synth.make_template(n,pre,FS); psd=synth.make_psd(n,FS); ft=build_filter(tmpl,psd,FS,pre)
x,_,_=synth.make_traces(b,tmpl,psd,FS,ft.ampres,seed=11,max_delay=min(2000,n//8)); x=x.astype(np.float32)
filt=orc.OFFilter(tmpl,psd,FS,pre); ref=orc.process_events(filt,x.astype(np.float64),'unconstrained')
plan=OFPlan(n,pre,FS,max_batch=4096,engine=sys.argv[1]); plan.set_filter(0,ft); sid=plan.add_search(0,'delay'); o=plan.search_offset(0,sid)
g=plan.process(torch.as_tensor(x,device='cuda')).cpu().numpy().astype(np.float64)
same=g[:,o+7].astype(np.int64)==ref['index']
amp=np.abs(g[:,o]-ref['amp'])/np.maximum(np.abs(ref['amp']),ft.ampres); chi=np.abs(g[:,o+2]-ref['chi2'])/np.abs(ref['chi2'])
print(sys.argv[1], 'flips',int((~same).sum()),'amp %.2e chi2 %.2e p999 %.2e'%(amp[same].max(),chi[same].max(),np.quantile(chi[same],0.999)))
