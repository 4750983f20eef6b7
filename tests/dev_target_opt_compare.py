"""Developer script: the in-kernel L-BFGS (host emulation of csrc/gp_target_fit.hip) against scipy L-BFGS-B on a REAL configs[4] refit
problem dumped on the GPU box by tools/dev_dump_target_problem.py (gpurun_out/tfp.npz: 80 target points, 32 sources, 5 start points).
Lives under tests/ because it uses the test-only host emulation.  Result of round 3: profiles/r03_notes.md."""
import ctypes, os, subprocess, sys, time
import numpy as np, scipy.optimize, torch
sys.path.insert(0, "/root/repo")
from tests._target_problem import TARGET_SPEC, pack_lower
ROOT="/root/repo"; CSRC=os.path.join(ROOT,"scalable-meta-learning-with-gaussian-processes_amd","csrc")
so="/tmp/target_fit_emul.so"
subprocess.run(["g++","-O2","-std=c++17","-fPIC","-shared","-x","c++","-I",CSRC,os.path.join(ROOT,"tests","host_emul","target_fit_emul.cpp"),"-o",so],check=True)
lib=ctypes.CDLL(so); dp,ip=ctypes.POINTER(ctypes.c_double),ctypes.POINTER(ctypes.c_int32)
lib.emul_target_fit.restype=ctypes.c_int
lib.emul_target_fit.argtypes=[dp,dp,dp,dp,ctypes.c_double,ctypes.c_double,dp,dp]+[ctypes.c_int]*8+[ctypes.c_double,ctypes.c_double,dp,dp,ip,dp,ip]
d=np.load("/root/repo/gpurun_out/tfp.npz")
n,T=d["means"].shape; D=d["X"].shape[1]
mt=np.ascontiguousarray(d["means"].T); covs=torch.from_numpy(d["covs"]).permute(2,0,1).contiguous()
cp=np.ascontiguousarray(pack_lower(covs.permute(1,2,0)).numpy()) if False else None
# pack lower (T, n(n+1)/2)
il=np.tril_indices(n); cp=np.ascontiguousarray(covs.numpy()[:,il[0],il[1]])
X=np.ascontiguousarray(d["X"]); y=np.ascontiguousarray(d["y"]); spec=np.array(TARGET_SPEC,dtype=np.float64)
# the model's real spec? (weights prior etc.) -- use the test spec's; what matters is iteration counts
P_=lambda a:a.ctypes.data_as(dp)
def call(z,mode,max_iter=200,history=10,kind=1):
    z=np.ascontiguousarray(np.atleast_2d(z).copy()); B,P=z.shape
    value,grad=np.zeros(B),np.zeros((B,P)); info,jit,stats=np.zeros(B,dtype=np.int32),np.zeros(B),np.zeros((B,4),dtype=np.int32)
    rc=lib.emul_target_fit(P_(mt),P_(cp),P_(X),P_(y),float(d["m"]),float(d["s"]),P_(spec),P_(z),B,n,T,D,kind,mode,max_iter,history,1e-5,2.2e-9,P_(value),P_(grad),info.ctypes.data_as(ip),P_(jit),stats.ctypes.data_as(ip))
    assert rc==0
    return value,grad,stats,z
z0=d["z0"]
for hist in (10,16):
  t0=time.time(); v,g,st,z=call(z0,1,200,hist); print("kernel L-BFGS hist",hist,": value",np.round(v,5),"its/evals/status",st[:,:3].tolist(), f"{time.time()-t0:.1f}s")
nev=[0]
def fun(zv):
    nev[0]+=1
    v,g,_,_=call(zv,0); return -v[0],-g[0]
bounds=[(None,None)]*(D+2)+[(1e-10,None)]*T
for b in range(z0.shape[0]):
    nev[0]=0
    r=scipy.optimize.minimize(fun,z0[b],jac=True,method="L-BFGS-B",bounds=bounds,options=dict(maxiter=200))
    print("scipy start",b,": value",round(-r.fun,5),"its",r.nit,"evals",nev[0],r.message)
