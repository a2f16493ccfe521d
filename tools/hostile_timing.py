import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import x265_amd
x265_amd.init(0)
rng=np.random.default_rng(0)
for name, gen in (("noise", lambda: rng.integers(0,256,(9,1088,1920),dtype=np.uint8)),
                  ("flat", lambda: np.full((9,1088,1920),77,np.uint8)),
                  ("tiny 64x64", lambda: rng.integers(0,256,(9,64,64),dtype=np.uint8))):
    fr=torch.from_numpy(gen()).cuda()
    for method in ("sea","brute"):
        out=None
        for it in range(3):
            torch.cuda.synchronize(); t0=time.perf_counter()
            out=x265_amd.me_fullsearch(fr[1:],fr[:-1],64,64,method=method,out=out)
            torch.cuda.synchronize(); dt=time.perf_counter()-t0
        a=x265_amd.mv_to_numpy(out)
        if method=="sea": keep=a
        else: assert (keep==a).all(), name
        print(name, method, "%.2f ms"%(dt*1e3))
