#!/usr/bin/env python3
"""The search stage's kernels with the reference planes tightly packed against planes inside Y,U,V pictures (frame stride
1.5 W H, the closed loop's layout).   python tools/box_stride.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import x265_amd  # noqa: E402
from x265_amd import api, synth  # noqa: E402

x265_amd.init(0)
lib = api._lib()
W, H, R, NF = 1920, 1088, 64, 16
seq = np.stack([synth.pad_plane(f, 64) for f in synth.make_sequence(W, 1080, NF + 1, search_range=R, seed=12345)])
ws = torch.empty(api.me_workspace_bytes(W, H, R, NF, 1), dtype=torch.uint8, device="cuda")
nctu = (W // 64) * (H // 64)
mv = torch.empty((NF, 1, nctu, 2), dtype=torch.int32, device="cuda")
for name, fs in (("packed", W * H), ("in Y,U,V pictures", W * H * 3 // 2), ("+ 256 bytes", W * H + 256), ("+ 4096 bytes", W * H + 4096)):
    cur = torch.zeros((NF, fs), dtype=torch.uint8, device="cuda")
    ref = torch.zeros((NF, fs), dtype=torch.uint8, device="cuda")
    cur[:, :W * H] = torch.from_numpy(seq[1:].reshape(NF, -1)).cuda()
    ref[:, :W * H] = torch.from_numpy(seq[:-1].reshape(NF, -1)).cuda()
    lib.x265amd_debug_me_timing(1)
    acc = np.zeros(4)
    buf = (ctypes.c_float * 4)()
    for it in range(24):
        st = lib.x265amd_me_fullsearch_sea(cur.data_ptr(), ref.data_ptr(), W, H, W, 64, R, NF, fs, fs, 1, 0, mv.data_ptr(),
                                           ws.data_ptr(), ws.numel(), None)
        assert st == 0, st
        torch.cuda.synchronize()
        lib.x265amd_debug_me_last_ms(buf)
        if it >= 8:
            acc += np.array(buf[:])
    lib.x265amd_debug_me_timing(0)
    print("%-20s frame stride %8d: box %.4f  bounds %.4f  exact %.4f ms" % (name, fs, acc[0] / 16, acc[2] / 16, acc[3] / 16), flush=True)
