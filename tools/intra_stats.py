"""Where the cycles of a CU go in the fused intra kernel (x265amd_debug_set_intra_stats).  python tools/intra_stats.py [batch]"""
import ctypes as C
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265_amd
from x265_amd import synth, api
x265_amd.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w, h, cu, qp = 1920, 1088, 32, 32
probe = synth.make_intra_probe(w, h, 8)
yuv = torch.from_numpy(np.ascontiguousarray(probe[[k % 8 for k in range(B)]])).cuda()
out = x265_amd.intra_encode_frames(yuv, w, h, cu, qp)
ncu = (w // cu) * (h // cu)
stats = torch.zeros((B, ncu, 16), dtype=torch.int32, device="cuda")
lib = api._lib()
lib.x265amd_debug_set_intra_stats.argtypes = [C.c_void_p]
lib.x265amd_debug_set_intra_stats(stats.data_ptr())
x265_amd.intra_encode_frames(yuv, w, h, cu, qp, out=out)
torch.cuda.synchronize()
lib.x265amd_debug_set_intra_stats(None)
s = stats.cpu().numpy().astype(np.int64).reshape(-1, 16) & 0xffffffff
names = ["loads -> LDS", "smooth/DC/MPM", "35 luma + 5 chroma SADs", "luma argmin", "luma pred store (+ chroma SAD of the luma mode)",
         "chroma argmin", "chroma pred store", "barrier before TQ", "TQ luma wave"]
d = (s[:, 1:10] - s[:, 0:9]) & 0xffffffff
tot = (s[:, 9] - s[:, 0]) & 0xffffffff
print(f"batch {B}: {s.shape[0]} CUs; cycles per CU, mean / median / p90")
for i, n in enumerate(names):
    print(f"  {n:50s} {d[:, i].mean():8.0f} {np.median(d[:, i]):8.0f} {np.percentile(d[:, i], 90):8.0f}")
print(f"  {'total (start .. luma TQ done)':50s} {tot.mean():8.0f} {np.median(tot):8.0f} {np.percentile(tot, 90):8.0f}")
