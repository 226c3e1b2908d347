import numpy as np, time, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from lightkurve_amd import _capi, synth
from oracle import np_oracle as O
t,y,e,off = synth.ls_batch(1, 64, 20000)
for nt in (1,2,3,4):
    _capi.ls_power_batch(t,y,off,f0=0.0036,df=0.0036,M=100000,normalization="lk_amplitude",nterms=nt)
    t0=time.perf_counter(); p=_capi.ls_power_batch(t,y,off,f0=0.0036,df=0.0036,M=100000,normalization="lk_amplitude",nterms=nt); dt=time.perf_counter()-t0
    print(nt, "%.1f ms"%(dt*1e3), "%.3g freq*targets/s (host API, PCIe incl.)"%(64*1e5/dt))
