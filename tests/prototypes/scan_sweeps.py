"""Numerics of the parallel-scan form of the block LDL^T sweeps (lbfgs_minco_persistent.h: chain_solve) against the node-by-node
walk, both in float64, referenced to a long-double walk: errors of the scan stay within a factor two of the walk."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.prototypes.proto_reduced import consts, assemble
def run(s,N,Tlo,Thi,seed,dtype=np.float64):
    rng=np.random.default_rng(seed)
    Bc,Mc=consts(s); m=s-1
    T=rng.uniform(Tlo,Thi,size=N)
    Ws,Kd,Ko,fixed=assemble(s,T,3,Mc)
    # pin fixed entries (c=3 -> first 2 of m at ends)
    Kd=Kd.copy();Ko=Ko.copy()
    for k in (0,N):
        for j in range(m):
            if fixed[k,j]:
                if k+1<=N and k<N: Ko[k][j,:]=0
                if k>0: Ko[k-1][:,j]=0
                Kd[k][:,j]=0;Kd[k][j,:]=0;Kd[k][j,j]=1
    r=rng.standard_normal((N+1,m))*np.array([1.0/ (T.mean()**(j)) for j in range(m)])
    def chain(dt):
        Kd_=Kd.astype(dt);Ko_=Ko.astype(dt);r_=r.astype(dt)
        S=[None]*(N+1);H=[None]*N;Si=[None]*(N+1)
        S[0]=Kd_[0]
        for k in range(N+1):
            if k>0: S[k]=Kd_[k]-Ko_[k-1].T@Si[k-1]@Ko_[k-1]
            Si[k]=np.linalg.inv(S[k].astype(np.float64)).astype(dt) if dt==np.float64 else np.array(np.linalg.inv(S[k].astype(np.float64)),dtype=dt)
            if k<N: H[k]=Si[k]@Ko_[k]
        return S,Si,H,r_
    S,Si,H,r_=chain(np.float64)
    # serial in longdouble using float64 factor (isolate sweep error)
    def serial(dt):
        Hh=[h.astype(dt) for h in H];Sii=[x.astype(dt) for x in Si];rr=r_.astype(dt)
        y=[None]*(N+1);y[0]=rr[0]
        for k in range(1,N+1): y[k]=rr[k]-Hh[k-1].T@y[k-1]
        z=[Sii[k]@y[k] for k in range(N+1)]
        x=[None]*(N+1);x[N]=z[N]
        for k in range(N-1,-1,-1): x[k]=z[k]-Hh[k]@x[k+1]
        return np.array(y),np.array(x)
    def scan():
        # forward: y_k = G_k y_{k-1} + r_k, G_k=-H_{k-1}^T, G_0=0
        M=np.zeros((N+1,m,m));v=r_.copy()
        for k in range(1,N+1): M[k]=-H[k-1].T
        d=1
        while d<=N:
            Mn=M.copy();vn=v.copy()
            for k in range(d,N+1):
                vn[k]=v[k]+M[k]@v[k-d]; Mn[k]=M[k]@M[k-d]
            M,v=Mn,vn; d*=2
        y=v
        z=np.array([Si[k]@y[k] for k in range(N+1)])
        M=np.zeros((N+1,m,m));v=z.copy()
        for k in range(N): M[k]=-H[k]
        d=1
        while d<=N:
            Mn=M.copy();vn=v.copy()
            for k in range(0,N+1-d):
                vn[k]=v[k]+M[k]@v[k+d]; Mn[k]=M[k]@M[k+d]
            M,v=Mn,vn; d*=2
        return y,v
    yl,xl=serial(np.longdouble)
    ys,xs=serial(np.float64)
    yc,xc=scan()
    sc=np.abs(xl).max(axis=0)
    return (np.abs(xs-xl)/sc).max(), (np.abs(xc-xl)/sc).max(), max(np.abs(h).max() for h in H)
for (s,N,lo,hi) in ((3,16,0.5,2.0),(4,8,0.5,2.0),(3,16,0.1,5.0),(4,16,0.1,5.0),(3,16,0.02,20.0),(4,8,0.02,20.0)):
    es=[];ec=[];hm=[]
    for seed in range(30):
        a,b,h=run(s,N,lo,hi,seed); es.append(a);ec.append(b);hm.append(h)
    print(s,N,lo,hi,"serial err max %.2e  scan err max %.2e  median scan %.2e  |H|max %.1e"%(max(es),max(ec),np.median(ec),max(hm)))
