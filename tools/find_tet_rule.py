#!/usr/bin/env python3
"""Numerical search for a fully symmetric, positive, interior quadrature rule of a given degree on the tetrahedron (round 4: the
171-point degree-13 rule of include/cfdh_quad_tet.h replaces the 343-point collapsed Gauss rule).

Unknowns: one weight per orbit and the orbit's free barycentric parameters (orbit types: centroid; (a,a,a,1-3a) 4 points; (a,a,b,b)
6 points; (a,a,b,1-2a-b) 12 points; (a,b,c,1-a-b-c) 24 points).  Equations: the orbit sums of the orthonormal
Proriol-Koornwinder-Dubiner basis up to the degree vanish (39 independent symmetric equations at degree 13, picked by a pivoted QR
on random orbits), the weights sum to 1.  Levenberg-Marquardt (scipy least_squares with bounds that keep the points inside and
the weights positive) from random starts, then a polish to 1e-15 and a check against ALL 560 basis functions.

  DEG=13 python tools/find_tet_rule.py 1,5,3,7,2 1 400      # orbit counts per type, seed, tries -> rule_d13_<...>.npy

tools/gen_quadrature_tet.py holds the generators found this way and writes the header (checking exactness once more, against the
collapsed Gauss-Jacobi rule)."""
import numpy as np, itertools, sys, math, os
from scipy.optimize import least_squares
from scipy.special import eval_jacobi, roots_jacobi
DEG=int(os.environ.get("DEG","13"))
PQR=[(p,q,r) for p in range(DEG+1) for q in range(DEG+1-p) for r in range(DEG+1-p-q)]
Pn=np.array([t[0] for t in PQR]);Qn=np.array([t[1] for t in PQR]);Rn=np.array([t[2] for t in PQR])
perms4=np.array(list(itertools.permutations(range(4))))
def psi(L, sel=None):
    """orthogonal basis at barycentric points L (n,4) -> (nfun, n)"""
    x=-1+2*L[:,1]; y=-1+2*L[:,2]; z=-1+2*L[:,3]
    a=2*(1+x)/(-y-z)-1; b=2*(1+y)/(1-z)-1; c=z
    P,Q,R=(Pn,Qn,Rn) if sel is None else (Pn[sel],Qn[sel],Rn[sel])
    f=eval_jacobi(P[:,None],0,0,a[None,:])*((1-b[None,:])/2)**P[:,None]*eval_jacobi(Q[:,None],2*P[:,None]+1,0,b[None,:])*((1-c[None,:])/2)**(P+Q)[:,None]*eval_jacobi(R[:,None],2*(P+Q)[:,None]+2,0,c[None,:])
    return f
def big_rule(m=16):
    # collapsed Gauss-Jacobi on the tet, barycentric points, weights sum 1
    x0,w0=roots_jacobi(m,0,0); x1,w1=roots_jacobi(m,1,0); x2,w2=roots_jacobi(m,2,0)
    pts=[];ws=[]
    for i in range(m):
        for j in range(m):
            for k in range(m):
                c=x2[i]; b=x1[j]; a=x0[k]
                z=c; y=(1+b)*(1-z)/2-1; x=(1+a)*(-y-z)/2-1
                l1=(x+1)/2;l2=(y+1)/2;l3=(z+1)/2
                pts.append([1-l1-l2-l3,l1,l2,l3]); ws.append(w0[k]*w1[j]*w2[i])
    ws=np.array(ws); return np.array(pts), ws/ws.sum()
BP,BW=big_rule()
F=psi(BP)
NRM=np.sqrt((F*F*BW).sum(axis=1))
assert abs((F[0]*F[5]*BW).sum())<1e-12
NP={0:1,1:4,2:6,3:12,4:24}; NPAR={0:0,1:1,2:1,3:2,4:3}
def base_points(x,struct):
    """returns bases (norb,4), weights-per-point (norb), multiplicity factor"""
    B=[];W=[];M=[];i=0
    for kind,n in enumerate(struct):
        for _ in range(n):
            w=x[i]; par=x[i+1:i+1+NPAR[kind]]; i+=1+NPAR[kind]
            if kind==0: b=[.25,.25,.25,.25]
            elif kind==1: a=par[0]; b=[a,a,a,1-3*a]
            elif kind==2: a=par[0]; b=[a,a,.5-a,.5-a]
            elif kind==3: a,bb=par; b=[a,a,bb,1-2*a-bb]
            else: a,bb,c=par; b=[a,bb,c,1-a-bb-c]
            B.append(b);W.append(w);M.append(NP[kind]/24.0)
    return np.array(B),np.array(W),np.array(M)
# choose independent symmetric equations
rng0=np.random.default_rng(0)
def sym_matrix(nsamp=200):
    cols=[]
    for _ in range(nsamp):
        p=rng0.dirichlet([1.0]*4)
        while p.min()<.02: p=rng0.dirichlet([1.0]*4)
        cols.append((psi(p[perms4])/NRM[:,None]).sum(axis=1))
    return np.array(cols).T   # nfun x nsamp
SM=sym_matrix()
from scipy.linalg import qr
_,_,piv=qr(SM.T,pivoting=True,mode='economic')  # columns = functions
sv=np.linalg.svd(SM,compute_uv=False)
RANK=int((sv>1e-9*sv[0]).sum())
SEL=np.sort(piv[:RANK])
if 0 not in SEL: SEL=np.sort(np.append(SEL,0))
def resid(x,struct):
    B,W,M=base_points(x,struct)
    P=B[:,perms4].reshape(-1,4)           # norb*24 x 4
    f=psi(P,SEL)/NRM[SEL][:,None]           # nsel x norb*24
    f=f.reshape(len(SEL),len(W),24).sum(axis=2)
    r=f@(W*M)
    r[SEL==0]-=1.0/NRM[0]
    return r
def bounds(struct):
    lo=[];hi=[]
    for kind,n in enumerate(struct):
        for _ in range(n):
            lo.append(1e-6);hi.append(1.0)
            if kind==1: lo.append(5e-3);hi.append(1/3-2e-3)
            if kind==2: lo.append(5e-3);hi.append(.5-5e-3)
            if kind==3: lo+= [5e-3,5e-3];hi+=[.5-5e-3,1-1e-2]
            if kind==4: lo+= [5e-3]*3;hi+=[1-1.5e-2]*3
    return np.array(lo),np.array(hi)
def start(struct,rng):
    x=[]
    npts=sum(NP[k]*n for k,n in enumerate(struct))
    for kind,n in enumerate(struct):
        for _ in range(n):
            x.append(rng.uniform(.3,1.5)/npts)
            if kind==1: x.append(rng.uniform(.02,.31))
            if kind==2: x.append(rng.uniform(.02,.48))
            if kind==3:
                while True:
                    a=rng.uniform(.02,.48);b=rng.uniform(.02,.9)
                    if 1-2*a-b>.02: break
                x+=[a,b]
            if kind==4:
                while True:
                    p=rng.dirichlet([1.5]*4)
                    if p.min()>.02: break
                x+=list(p[:3])
    return np.array(x)
def interior(x,struct):
    B,W,M=base_points(x,struct)
    return W.min()>0 and B.min()>2e-3
def full_check(x,struct):
    B,W,M=base_points(x,struct)
    P=B[:,perms4].reshape(-1,4)
    f=(psi(P)/NRM[:,None]).reshape(len(PQR),len(W),24).sum(axis=2)@(W*M)
    f[0]-=1/NRM[0]
    return np.abs(f).max()
if __name__=="__main__":
    struct=tuple(int(v) for v in sys.argv[1].split(','))
    seed=int(sys.argv[2]); tries=int(sys.argv[3])
    rng=np.random.default_rng(seed)
    lo,hi=bounds(struct)
    npts=sum(NP[k]*n for k,n in enumerate(struct))
    print('deg',DEG,'struct',struct,'points',npts,'unknowns',len(lo),'sym equations',len(SEL),'rank',RANK,flush=True)
    for t in range(tries):
        x0=np.clip(start(struct,rng),lo+1e-9,hi-1e-9)
        s=least_squares(resid,x0,args=(struct,),bounds=(lo,hi),xtol=1e-14,ftol=1e-14,gtol=1e-14,max_nfev=300)
        rn=np.abs(s.fun).max()
        if rn<1e-7:
            s=least_squares(resid,s.x,args=(struct,),bounds=(lo,hi),xtol=3e-16,ftol=1e-30,gtol=1e-30,max_nfev=1000)
            rn=np.abs(s.fun).max()
        ok=rn<5e-14 and interior(s.x,struct)
        print(t,'res %.2e'%rn,'nfev',s.nfev,('ok full %.2e'%full_check(s.x,struct)) if ok else '',flush=True)
        if ok:
            np.save('rule_d%d_%s_%d.npy'%(DEG,sys.argv[1].replace(',','-'),seed),s.x)
            break
