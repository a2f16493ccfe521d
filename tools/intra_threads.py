"""Time of x265amd_intra_encode_frames by workgroup size of the step kernel (x265amd_debug_intra_threads).  python tools/intra_threads.py"""
import ctypes as C
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265_amd
from x265_amd import synth, api
x265_amd.init(0)
w, h, cu, qp = 1920, 1088, 32, 32
probe = synth.make_intra_probe(w, h, 8)
lib = api._lib()
lib.x265amd_debug_intra_threads.argtypes = [C.c_int]
for B in (1, 4, 8, 16, 32):
    yuv = torch.from_numpy(np.ascontiguousarray(probe[[k % 8 for k in range(B)]])).cuda()
    out = x265_amd.intra_encode_frames(yuv, w, h, cu, qp)
    res = []
    for th in (0, 256, 512, 1024):
        lib.x265amd_debug_intra_threads(th)
        for _ in range(2):
            x265_amd.intra_encode_frames(yuv, w, h, cu, qp, out=out)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            x265_amd.intra_encode_frames(yuv, w, h, cu, qp, out=out)
        ev[1].record()
        torch.cuda.synchronize()
        res.append(ev[0].elapsed_time(ev[1]) / 5)
    lib.x265amd_debug_intra_threads(0)
    print(f"batch {B:2d}: auto {res[0]:.3f} ms | 256: {res[1]:.3f} | 512: {res[2]:.3f} | 1024: {res[3]:.3f}")
