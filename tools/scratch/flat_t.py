import sys, time, numpy as np
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import torch
from lightkurve_amd import _capi, synth
B=256; N=20000
t,y,_ = synth.ls_batch(1,B,N)[:3] if False else (None,None,None)
rng=np.random.default_rng(0)
ts=[];fs=[]
for b in range(B):
    tt=synth.tess_like_times(np.random.default_rng(b),N)
    ts.append(tt); fs.append(1+1e-3*np.sin(2*np.pi*tt/3.0)+5e-4*rng.standard_normal(N))
n_off=np.arange(B+1,dtype=np.int64)*N
t=np.concatenate(ts); f=np.concatenate(fs)
for (w,ni) in ((401,1),(401,2),(401,3),(101,3),(11,3)):
    for rep in range(2):
        t0=time.perf_counter()
        tr=_capi.savgol_trend_batch(t,f,n_off,None,w,2,5,ni,3)
        dt=time.perf_counter()-t0
    print(w,ni,"%.2f ms"%(dt*1e3))
