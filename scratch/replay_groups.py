import numpy as np
rng=np.random.default_rng(1)
n=200000
P=rng.random((n,3)); Q=rng.random((n,3))
G=int(round((n/2)**(1/3))); h=1.0/G
def cells(X): return np.minimum((X/h).astype(int),G-1)
cp=cells(P)
# snake irrelevant: use plain row-major with x fastest; runs along x contiguous
key=(cp[:,2]*G+cp[:,1])*G+cp[:,0]
order=np.argsort(key,kind='stable'); P=P[order]; key=key[order]
start=np.searchsorted(key,np.arange(G*G*G+1))
# queries sorted by their cell
cq=cells(Q); kq=(cq[:,2]*G+cq[:,1])*G+cq[:,0]; oq=np.argsort(kq,kind='stable'); Q=Q[oq]; cq=cq[oq]
rows=[(0,0),(-1,0),(1,0),(0,-1),(0,1),(-1,-1),(1,-1),(-1,1),(1,1)]
def groups_for(qi, sort_mode):
    q=Q[qi]; cx,cy,cz=cq[qi]
    def run(cy_,cz_,x0,x1):
        if cy_<0 or cy_>=G or cz_<0 or cz_>=G: return None
        x0=max(x0,0); x1=min(x1,G-1)
        if x1<x0: return None
        b=(cz_*G+cy_)*G
        return start[b+x0], start[b+x1+1]
    def scan(s,e,best):
        ng=0
        p=s
        while p<e:
            d=((P[p:p+4]-q)**2).sum(1)   # may run past e (like the kernel)
            best=min(best,d.min()); p+=4; ng+=1
        return best,ng
    s,e=run(cy,cz,cx-1,cx+1)
    best,g0=scan(s,e,np.inf)
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
    if sort_mode: ent.sort(key=lambda t:t[0])
    g=0
    for rlb,(s,e) in ent:
        if best<rlb: continue
        best,ng=scan(s,e,best); g+=ng
    return g0,g
nq=64*60
base=64*1000
for mode in (0,1):
    res=np.array([groups_for(base+i,mode) for i in range(nq)])
    c=res[:,0].reshape(-1,64); o=res[:,1].reshape(-1,64)
    print("sorted" if mode else "fixed ", "centre mean %.2f max/wave %.2f | outer mean %.2f max/wave %.2f"%(c.mean(),c.max(1).mean(),o.mean(),o.max(1).mean()))
