"""Per-CTU statistics of the successive-elimination search (debug aid)."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265_amd
from x265_amd import synth, api
x265_amd.init(0)
W, H, R = 1920, 1088, 64
seq = np.stack([synth.pad_plane(f, 64) for f in synth.make_sequence(1920, 1080, 3, search_range=R, seed=12345)])
fr = torch.from_numpy(seq).cuda()
nctu = (W // 64) * (H // 64)
stats = torch.zeros((2, nctu, 8), dtype=torch.int32, device="cuda")
lib = api._lib()
lib.x265amd_debug_set_me_stats.argtypes = [C.c_void_p]
lib.x265amd_debug_set_me_stats(stats.data_ptr())
mv = x265_amd.me_fullsearch(fr[1:], fr[:-1], 64, R, method="sea")
torch.cuda.synchronize()
lib.x265amd_debug_set_me_stats(None)
s = stats.cpu().numpy().astype(np.int64)[0]
names = ["n_pass0", "n_pass1"]
for i, n in enumerate(names):
    print(f"{n:14s} mean {s[:, i].mean():10.1f}  median {np.median(s[:, i]):10.1f}  max {s[:, i].max():10d}")
interior = [i for i in range(nctu) if 2 <= i % 30 <= 27 and 2 <= i // 30 <= 14]
print("interior CTUs:", {n: float(s[interior, i].mean()) for i, n in enumerate(names)})
