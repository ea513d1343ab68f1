import numpy as np
rng=np.random.default_rng(1)
n=200000
P=rng.random((n,3)); Q=rng.random((n,3))
G=int(round((n/2)**(1/3))); h=1.0/G
def cells(X): return np.minimum((X/h).astype(int),G-1)
cp=cells(P)
key=(cp[:,2]*G+cp[:,1])*G+cp[:,0]
order=np.argsort(key,kind='stable'); P=P[order]; key=key[order]
start=np.searchsorted(key,np.arange(G*G*G+1))
cq=cells(Q); kq=(cq[:,2]*G+cq[:,1])*G+cq[:,0]; oq=np.argsort(kq,kind='stable'); Q=Q[oq]; cq=cq[oq]
rows=[(0,0),(-1,0),(1,0),(0,-1),(0,1),(-1,-1),(1,-1),(-1,1),(1,1)]
def lane(qi):
    """returns centre groups, list of per-group events for the adaptive scan: sequence of (rlb, ngroups) runs and the best after each group"""
    q=Q[qi]; cx,cy,cz=cq[qi]
    def run(cy_,cz_,x0,x1):
        if cy_<0 or cy_>=G or cz_<0 or cz_>=G: return None
        x0=max(x0,0); x1=min(x1,G-1)
        if x1<x0: return None
        b=(cz_*G+cy_)*G
        return start[b+x0], start[b+x1+1]
    s,e=run(cy,cz,cx-1,cx+1)
    best=np.inf; g0=0; p=s
    while p<e:
        d=((P[p:p+4]-q)**2).sum(1); best=min(best,d.min()); p+=4; g0+=1
    mxl=q[0]-cx*h; mxh=(cx+1)*h-q[0]
    my={-1:q[1]-cy*h,0:0.0,1:(cy+1)*h-q[1]}; mz={-1:q[2]-cz*h,0:0.0,1:(cz+1)*h-q[2]}
    ent=[]
    for (oy,oz) in rows[1:]:
        rlb=my[oy]**2+mz[oz]**2
        if best<rlb: continue
        x0=cx-1 if not (best<mxl**2+rlb) else cx
        x1=cx+1 if not (best<mxh**2+rlb) else cx
        r=run(cy+oy,cz+oz,x0,x1)
        if r is None or r[1]<=r[0]: continue
        ent.append((rlb,r))
    return q,g0,best,ent
def simulate(qi,T):
    q,g0,best,ent=lane(qi)
    done=0; i=0; p=None; e=None
    # adaptive own scan for T groups
    while done<T:
        if p is None or p>=e:
            while i<len(ent) and best<ent[i][0]: i+=1
            if i>=len(ent): return g0,done,0
            p,e=ent[i][1]; i+=1
        d=((P[p:p+4]-q)**2).sum(1); best=min(best,d.min()); p+=4; done+=1
    rem=0
    if p is not None and p<e: rem+=(e-p+3)//4
    for j in range(i,len(ent)):
        if not (best<ent[j][0]): rem+=(ent[j][1][1]-ent[j][1][0]+3)//4
    return g0,done,rem
nw=40; base=64*1000
for T in (1000,2,3,4,5,6):
    res=np.array([simulate(base+i,T) for i in range(64*nw)])
    own=res[:,1].reshape(nw,64); rem=res[:,2].reshape(nw,64)
    trips=own.max(1)+np.ceil(rem.sum(1)/64)
    print("T=%4d  own max/wave %.2f  W mean %.1f (max %d)  dealt trips %.2f  total outer trips %.2f"%(T,own.max(1).mean(),rem.sum(1).mean(),rem.sum(1).max(),np.ceil(rem.sum(1)/64).mean(),trips.mean()))
