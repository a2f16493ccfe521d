import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import x265_amd
from x265_amd import synth
x265_amd.init(0)
W,H,R,NF=1920,1088,64,8
seq=np.stack([synth.pad_plane(f,64) for f in synth.make_sequence(1920,1080,NF+1,search_range=R,seed=12345)])
fr=torch.from_numpy(seq).cuda()
for ctu in (64,32,16):
    for method in ("brute","basic"):
        out=None
        for it in range(3):
            torch.cuda.synchronize(); t0=time.perf_counter()
            out=x265_amd.me_fullsearch(fr[1:],fr[:-1],ctu,R,method=method,out=out)
            torch.cuda.synchronize(); dt=time.perf_counter()-t0
        print(ctu, method, "%.2f ms  %.0f Mpix/s"%(dt*1e3, 1920*1080*NF/dt/1e6))
