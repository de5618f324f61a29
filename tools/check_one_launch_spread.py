#!/usr/bin/env python3
"""One-launch (scanned sweeps) against lockstep (walked sweeps) MINCO L-BFGS after one iteration, durations spread over up to
three decades: identical counters, costs equal to the accuracy envelope of the reduced system.
    gpurun -- 'python tools/check_one_launch_spread.py'"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import allocnet_amd as aa
from allocnet_amd.synth import corridor_problem
ctx=aa.Context(0)
for (s,c,N,M) in ((3,3,16,8),(4,3,8,8),(4,4,16,6),(3,3,5,6)):
    for lo,hi in ((0.5,2.0),(0.05,5.0),(0.02,20.0)):
        rng=np.random.default_rng(5)
        B=200
        head,tail,wps,T,hp=corridor_problem(rng,B,N,c,M)
        T=np.exp(rng.uniform(np.log(lo),np.log(hi),size=T.shape))
        pen=aa.make_penalty(rho=20.0,w_corridor=1e3,w_vel=1e2,w_acc=1e2,smooth_mu=1e-2,max_vel=3.0,max_acc=4.0,res=8,poly_rows=M)
        prm=aa.lbfgs_parameter_t(max_iterations=1)
        a=aa.lbfgs_minco(head,tail,wps,T,s,hpolys=hp,penalty=pen,param=prm,opt=3,ctx=ctx)
        b=aa.lbfgs_minco(head,tail,wps,T,s,hpolys=hp,penalty=pen,param=prm,opt=3|aa.lbfgs.OPT_LOCKSTEP,ctx=ctx)
        same=(a["status"]==b["status"])&(a["evals"]==b["evals"])
        rel=np.abs(a["cost"]-b["cost"])/np.abs(b["cost"])
        print(s,N,lo,hi,"same %.3f  cost rel diff max %.2e median %.2e  (same only max %.2e)"%(same.mean(),rel.max(),np.median(rel),rel[same].max()))
