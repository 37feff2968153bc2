import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import datagen
MUL=0x9D6EF916
def hashes(q): return ((q.astype(np.uint64)*MUL)&0xFFFFFFFF)>>16

def cheetah_flags(q):
    """sequential truth: flags (3=pred) for quads q (no raw copy handling)"""
    n=len(q); h=hashes(q).astype(np.int64)
    pred={}; A={}; B={}
    flags=np.zeros(n,np.uint8); last=0
    for i in range(n):
        qi=int(q[i]); hi=int(h[i])
        if pred.get(last,0)==qi:
            flags[i]=3
        else:
            a=A.get(hi,0); b=B.get(hi,0)
            if a==qi: flags[i]=1
            else:
                flags[i]=2 if b==qi else 0
                B[hi]=a; A[hi]=qi
            pred[last]=qi
        last=hi
    return flags,h

def prevctx(c):
    """for each step i: latest s<i with c[s]==c[i], else -1  (c: contexts array len n)"""
    n=len(c)
    order=np.lexsort((np.arange(n),c))
    cs=c[order]
    prev=np.full(n,-1,np.int64)
    same=cs[1:]==cs[:-1]
    prev[order[1:][same]]=order[:-1][same]
    return prev

def solve(flags,h_true,q_true,verbose=True):
    n=len(flags)
    pred=flags==3
    # contexts c[i]=h[i-1], c[0]=0.  known where i-1 is non-pred
    UNK=-1
    c=np.full(n+1,UNK,np.int64); c[0]=0
    c[1:][~pred]=h_true[~pred]
    c_true=np.concatenate([[0],h_true])
    # run starts: pred i with known c[i]
    # iteration 0: src for steps with known context over known-context steps only; constant offset continuation
    src=np.full(n,-1,np.int64)
    it=0
    idx=np.arange(n)
    # unique placeholder contexts for unknown (negative, distinct)
    def with_placeholders(c):
        cc=c[:n].copy()
        unk=cc==UNK
        cc[unk]=-(idx[unk]+2)
        return cc
    # bootstrap
    cc=with_placeholders(c)
    p=prevctx(cc)
    known_ctx=c[:n]!=UNK
    # src for pred steps with known ctx
    anchor=pred&known_ctx
    src[anchor]=p[anchor]
    # constant offset continuation: for pred steps with unknown ctx: offset of the nearest earlier anchor
    off=np.where(anchor,idx-src,0)
    # propagate: last anchor index
    last_anchor=np.maximum.accumulate(np.where(anchor,idx,-1))
    cont=pred&~anchor
    # anchors with src -1 (no previous): value 0 -> mark src=-1 (root zero)
    src[cont]=idx[cont]-(last_anchor[cont]-src[last_anchor[cont]])
    bad0=cont&(src[last_anchor]== -1)
    src[bad0]=-1
    iters=0
    while True:
        iters+=1
        # resolve contexts by pointer jumping: c[i+1] = c[src[i]+1] for pred i ; roots: known
        c2=np.full(n+1,UNK,np.int64); c2[0]=0; c2[1:][~pred]=h_true[~pred]
        ptr=np.arange(n+1)
        pi=idx[pred]
        ptr[pi+1]=np.where(src[pi]>=0,src[pi]+1,n+2)  # n+2: zero root
        # pointer jumping
        rounds=0
        ptr_ext=np.concatenate([ptr,[n+1,n+2]]) # sentinels
        cval=np.concatenate([c2,[UNK,0]])  # hash(0)=0 for zero root -> context after zero quad = hash(0)=0
        isroot=np.concatenate([(c2!=UNK),[True,True]])
        cur=ptr_ext.copy()
        while True:
            nr=~isroot[cur]
            if not nr.any(): break
            cur[nr]=cur[cur[nr]] if False else ptr_ext[cur[nr]]
            rounds+=1
            if rounds>n+5: raise RuntimeError('cycle')
        # (that loop is linear-chasing per round not doubling; count doubling rounds separately)
        cnew=cval[cur][:n+1]
        # exact prevctx on candidate contexts
        p=prevctx(cnew[:n])
        mism=pred&(p!=src)
        nm=int(mism.sum())
        first=int(np.argmax(mism)) if nm else n
        wrongc=int((cnew!=c_true).sum())
        if verbose: print(f"  iter {iters}: mismatches {nm}, first at {first}, wrong contexts {wrongc}, chase depth {rounds}")
        if nm==0:
            assert wrongc==0
            break
        src=np.where(pred,p,src)
        if iters>60: break
    return iters

if __name__=="__main__":
    kind=sys.argv[1] if len(sys.argv)>1 else 'prose'
    n=int(sys.argv[2]) if len(sys.argv)>2 else 1<<20
    data=datagen.by_kind(kind,n,seed=7) if kind!='rep' else datagen.rep_text(n,period=100003)
    q=data.view('<u4')
    flags,h=cheetah_flags(q)
    print(kind,n,"flag mix",np.bincount(flags,minlength=4)/len(flags))
    # run lengths
    pr=flags==3
    d=np.diff(np.concatenate([[0],pr.astype(np.int8),[0]]))
    starts=np.flatnonzero(d==1); ends=np.flatnonzero(d==-1); L=ends-starts
    print(" runs",len(L),"mean",L.mean() if len(L) else 0,"max",L.max() if len(L) else 0, "p99",np.percentile(L,99) if len(L) else 0)
    it=solve(flags,h,q)
    print(" iterations",it)
