import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import datagen
from ctx import cheetah_flags
def analyse(kind,n,W):
    data=datagen.by_kind(kind,n,seed=7) if kind!='rep' else datagen.rep_text(n,period=100003)
    q=data.view('<u4'); flags,h=cheetah_flags(q)
    nst=len(q); pred=flags==3
    c=np.concatenate([[0],h[:-1]])          # context of step i
    # depth: 0 if previous step non-pred (or i==0), else depth(prev)+1 where prev pred
    depth=np.zeros(nst,np.int32)
    run=0
    for i in range(nst):
        if i>0 and pred[i-1]:
            run+=1
        else:
            run=0
        depth[i]=run
    iswrite=~pred
    nwin=nst//W; bad=0; phases=[]
    for w in range(nwin):
        s=slice(w*W,(w+1)*W)
        cc=c[s]; dd=depth[s]; ww=iswrite[s]
        phases.append(dd.max()+1)
        dyn=np.flatnonzero(dd>0)
        viol=False
        for k in dyn:
            same=np.flatnonzero((cc[k+1:]==cc[k]))+k+1
            if len(same)==0: continue
            m=(dd[same]<dd[k])&(ww[same]|ww[k])
            if m.any(): viol=True;break
        bad+=viol
    print(f"{kind} W={W}: windows {nwin}, with violation {bad} ({100*bad/nwin:.1f}%), mean phases {np.mean(phases):.2f}, max {max(phases)}, dyn frac {np.mean(depth>0):.3f}")
for kind in ('prose','rep','binaryish','mixed'):
    for W in (64,256,1024):
        analyse(kind,1<<19,W)
