"""numpy prototype of density_amd/csrc/stream_parse.hip (record boundaries of a calm Chameleon stream from per-window tables), checked against
an FSM walk of the oracle's streams.  CPU only: python tools/parse_prototype.py"""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import datagen
from oracle import pyoracle
W=8192; NC=W//2; END=254
def popc64(a): return np.array([bin(int(x)).count('1') for x in a],dtype=np.int64)
def serial_head(enc):
    """true FSM walk until calm; returns (pos, block, index list)"""
    E=len(enc); pos=0; b=0; idx=[]
    pen=0; start=1; prev=0; counter=0
    while pos<E:
        # block_is_copy
        if (counter&15)==0 and start>1: start>>=1
        counter+=1
        if pen>0:
            idx.append(0x80); pos+=256; pen-=1
            if pen==0: start+=1
        else:
            if pos+8>E: break
            sig=int.from_bytes(enc[pos:pos+8],'little'); pc=bin(sig).count('1'); rl=264-2*pc
            if pos+rl>E: break
            idx.append(pc); inc= rl>=256
            if inc and prev: pen=start
            prev=1 if inc else 0
            pos+=rl
        b+=1
        if pen==0 and prev==0 and start==1 and b>=2: return pos,b,idx
        if b>4096: return None
    return None
def parse(enc):
    E=len(enc)
    h=serial_head(enc)
    if h is None: return None
    p0,b0,idx=h
    buf=np.frombuffer(enc+b'\0'*16,dtype=np.uint8)
    nW=(E-p0+W-1)//W
    T=np.zeros((nW,132),np.int64); C=np.zeros((nW,132),np.int64)
    allex=[]; 
    for w in range(nW):
        ws=p0+w*W
        c=np.arange(NC); p=ws+2*c
        # sig at each candidate
        ok=p+8<=E
        sig=np.zeros(NC,np.uint64)
        pp=np.minimum(p,E) 
        b8=np.stack([buf[pp+i].astype(np.uint64)<<np.uint64(8*i) for i in range(8)]).sum(0).astype(np.uint64)
        pc=np.unpackbits(b8.view(np.uint8).reshape(-1,8),axis=1).sum(1).astype(np.int64)
        rl=264-2*pc
        full=ok&(p+rl<=E)
        nxt=c+132-pc
        ex=np.full(NC,END,np.int64); cn=np.zeros(NC,np.int64)
        for g in range(NC//64-1,-1,-1):
            s=slice(64*g,64*g+64)
            n=nxt[s]; f=full[s]
            out=n>=NC
            e=np.where(out,n-NC,0); k=np.ones(64,np.int64)
            inn=~out
            nn=np.where(inn,n,0)
            e=np.where(inn,ex[nn],e); k=np.where(inn,1+cn[nn],k)
            ex[s]=np.where(f,e,END); cn[s]=np.where(f,k,0)
        T[w]=ex[:132]; C[w]=cn[:132]
        allex.append((ex,cn))
    # compose
    x=0; base=b0; ent=[]; bases=[]
    for w in range(nW):
        ent.append(x); bases.append(base)
        if x==END: continue
        base+=C[w][x]; x=T[w][x]
    total=base
    # emit
    index=list(idx)+[None]*(total-b0)
    endpos=None
    for w in range(nW):
        x=ent[w]
        if x==END: break
        c=x; b=bases[w]; ws=p0+w*W
        while c<NC:
            p=ws+2*c
            if p+8>E: endpos=p; break
            sig=int.from_bytes(enc[p:p+8],'little'); pc=bin(sig).count('1'); rl=264-2*pc
            if p+rl>E: endpos=p; break
            index[b]=pc; b+=1; c+=rl//2
        if endpos is not None: break
    if endpos is None: endpos=E  # ended exactly at a window boundary chain
    # pair check
    inc=[ (v is not None and v<0x80 and v<=4) for v in index]
    for i in range(max(b0-1,0),len(index)-1):
        if inc[i] and inc[i+1]: return ('fallback',i)
    return p0,b0,total,endpos,index
def truth(enc,n):
    """walk with true FSM"""
    E=len(enc); pos=0; idx=[]
    pen=0; start=1; prev=0; counter=0; b=0
    nfull=n//256
    while b<nfull:
        if (counter&15)==0 and start>1: start>>=1
        counter+=1
        if pen>0:
            idx.append(0x80); pos+=256; pen-=1
            if pen==0: start+=1
        else:
            sig=int.from_bytes(enc[pos:pos+8],'little'); pc=bin(sig).count('1'); rl=264-2*pc
            idx.append(pc); inc=rl>=256
            if inc and prev: pen=start
            prev=1 if inc else 0
            pos+=rl
        b+=1
    return idx,pos
if __name__ == '__main__':
  for kind,n in [("prose",3*1024*1024+77),("rep",5*1024*1024),("prose",1<<20)]:
      data=datagen.by_kind(kind,n,seed=5)
      enc=pyoracle.encode("chameleon",data)
      r=parse(enc)
      ti,tpos=truth(enc,n)
      if r is None or r[0]=='fallback': print(kind,n,'->',r); continue
      p0,b0,total,endpos,index=r
      print(kind,n,'p0',p0,'b0',b0,'total',total,'true blocks',len(ti),'endpos',endpos,'true end of full blocks',tpos,'index equal',index==ti)
