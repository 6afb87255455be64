#!/usr/bin/env python3
"""Digest of a wave timeline recorded by tools/trace_waves.py (-DRK_TRACE build): duration statistics, resident waves over
time (40 slices), per-XCD and per-class summaries, work / duration. usage: trace_digest.py <trace.npz>"""
import numpy as np, sys
d=np.load(sys.argv[1])
t0=d['t0'].astype(np.int64); t1=d['t1'].astype(np.int64); ok=t1>0
T=d['T'][ok]; R=d['R'][ok]; xcc=d['xcc'][ok]; hw=d['hw'][ok]; work=d['work'][ok].astype(np.float64)
t0=t0[ok]; t1=t1[ok]; base=t0.min(); t0=(t0-base)/100.0; t1=(t1-base)/100.0  # us
dur=t1-t0
if 'cyc' in d and d['cyc'][ok].max()>0:
    cyc=d['cyc'][ok].astype(np.float64); ghz=cyc/(dur*1e3)
    print('shader clock while the waves ran (cycles / wall time): median %.3f GHz, p10 %.3f, p90 %.3f'%(np.median(ghz),np.percentile(ghz,10),np.percentile(ghz,90)))
print('waves',ok.sum(),'span us',t1.max(),'kernel_ms',d['kernel_ms'][-3:])
print('dur us: mean %.1f med %.1f p90 %.1f p99 %.1f max %.1f'%(dur.mean(),np.median(dur),np.percentile(dur,90),np.percentile(dur,99),dur.max()))
# occupancy over time
edges=np.linspace(0,t1.max(),41)
occ=[]
for a,b in zip(edges[:-1],edges[1:]):
    occ.append(np.sum(np.clip(np.minimum(t1,b)-np.maximum(t0,a),0,None))/(b-a))
print('resident waves per 2.5% slice:', ' '.join('%d'%o for o in occ))
# per XCD end time and busy
for x in range(8):
    s=xcc==x
    print('xcd',x,'waves',s.sum(),'last end %.0f us'%t1[s].max(),'sum dur %.0f ms'%(dur[s].sum()/1e3), 'work %.3g'%work[s].sum())
# per R class
for r in sorted(set(R)):
    s=R==r
    print('R',r,'n',s.sum(),'first start %.0f last end %.0f mean dur %.1f work/dur %.3g'%(t0[s].min(),t1[s].max(),dur[s].mean(), work[s].sum()/dur[s].sum()))
# correlation of duration with work
print('corr(dur,work)=%.3f'%np.corrcoef(dur,work)[0,1])
# efficiency: work per wave-us as function of occupancy? rate in first half vs last 20%
mid=(t0+t1)/2
for a,b in ((0,0.5),(0.5,0.8),(0.8,0.9),(0.9,1.0)):
    s=(mid>=a*t1.max())&(mid<b*t1.max())
    print('waves centred in [%.1f,%.1f): n %d, work/dur %.3g'%(a,b,s.sum(), work[s].sum()/dur[s].sum()))
# late starters
order=np.argsort(-t1)[:10]
for i in order: print('late end: start %.0f end %.0f dur %.0f T %d R %d work %.3g xcd %d'%(t0[i],t1[i],dur[i],T[i],R[i],work[i],xcc[i]))
# start-time histogram (waves started per 5 % slice of the span) per class: how fast the launch fills the device
edges=np.linspace(0,t1.max(),21)
for r in sorted(set(R)):
    s=R==r
    h,_=np.histogram(t0[s],bins=edges)
    print('R',r,'starts per 5% slice:',' '.join('%d'%v for v in h))
