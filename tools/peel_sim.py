"""Rounds and single steps of peel_largest() (csrc/peel.hpp) on synthetic power spectra: python tools/peel_sim.py
Lane-sorted columns of 1025 / 2049 values, K = 65 taken; D = look-ahead depth of the first attempt per round."""
import numpy as np
rng=np.random.default_rng(0)
def simulate(p, K, D):
    pad=np.full((64,D+2),-1.0)
    P=np.concatenate([p,pad],1)
    taken=np.zeros(64,int); r=K; wm=0; tail=0; rounds=0
    ar=np.arange(64)
    while r>0:
        rounds+=1
        done=False
        for d in range(D,0,-1):
            T=P[ar,taken+d].max(); wm+=1
            cnt=np.zeros(64,int)
            for j in range(d): cnt+=(P[ar,taken+j]>=T)
            c=cnt.sum()
            if c<=r:
                taken+=cnt; r-=c; done=True; break
        if not done:
            tail=r
            cand=[(P[l,taken[l]],l) for l in range(64) if cnt[l]]
            cand.sort(reverse=True)
            for v,l in cand[:r]: taken[l]+=1
            r=0
    return rounds,wm,tail,taken
def check(vals,K,layout,D):
    n=len(vals)
    if layout=='strided':
        depth=(n+63)//64
        a=np.full(64*depth,-1.0); a[:n]=vals; p=a.reshape(depth,64).T
    else:
        depth2=(n+127)//128
        a=np.full(128*depth2,-1.0); a[:n]=vals; p=a.reshape(depth2,64,2).transpose(1,0,2).reshape(64,-1)
    p=-np.sort(-p,axis=1)
    rounds,wm,tail,taken=simulate(p,K,D)
    low=sum(p[l,taken[l]:][p[l,taken[l]:]>=0].sum() for l in range(64))
    ref=np.sort(vals)[:n-K].sum()
    assert abs(low-ref)<1e-9*max(1,ref),(low,ref)
    return rounds,wm,tail
for name,gen in [('noise',lambda n: rng.exponential(size=n)),
                 ('lobe',lambda n: np.exp(-np.arange(n)/8.0)+1e-6*rng.exponential(size=n)),
                 ('lobe+noise',lambda n: 100*np.exp(-np.arange(n)/10.0)+rng.exponential(size=n)),
                 ('peaks',lambda n: rng.exponential(size=n)+50*(np.arange(n)%97<3))]:
    for n,layout in [(1025,'strided'),(2049,'pairs'),(2049,'strided')]:
        for D in (2,3):
            res=np.array([check(gen(n),65,layout,D) for _ in range(50)])
            print(name,n,layout,'D',D,'rounds %.1f/%d wavemax %.1f/%d tail %.1f/%d'%(res[:,0].mean(),res[:,0].max(),res[:,1].mean(),res[:,1].max(),res[:,2].mean(),res[:,2].max()))
