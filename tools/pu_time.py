import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import x265_amd
from x265_amd import synth
x265_amd.init(0)
R, NF = 64, 8
seq = np.stack([synth.pad_plane(f, 64) for f in synth.make_sequence(1920, 1080, NF + 1, search_range=R, seed=12345)])
fr = torch.from_numpy(seq).cuda()
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = x265_amd.me_fullsearch_pu_all(fr[1:], fr[:-1], R)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("pu_all %.3f ms" % (dt * 1e3))
