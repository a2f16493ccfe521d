#!/usr/bin/env python3
"""Kernel split of the elimination search at 1080p +-64 x 16 frames for blocks of 64 and of 32 (the reference's default CU
size).   python tools/b32_time.py"""
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
cur = torch.from_numpy(seq[1:]).cuda()
ref = torch.from_numpy(seq[:-1]).cuda()
for B in (64, 32):
    ws = torch.empty(api.me_workspace_bytes(W, H, R, NF, 1, B), dtype=torch.uint8, device="cuda")
    mv = torch.empty((NF, 1, (W // B) * (H // B), 2), dtype=torch.int32, device="cuda")
    lib.x265amd_debug_me_timing(1)
    buf = (ctypes.c_float * 4)()
    acc = np.zeros(4)
    for it in range(12):
        st = lib.x265amd_me_fullsearch_sea(cur.data_ptr(), ref.data_ptr(), W, H, W, B, R, NF, W * H, W * H, 1, 0, mv.data_ptr(),
                                           ws.data_ptr(), ws.numel(), None)
        assert st == 0, st
        torch.cuda.synchronize()
        lib.x265amd_debug_me_last_ms(buf)
        if it >= 4:
            acc += np.array(buf[:])
    lib.x265amd_debug_me_timing(0)
    brute = x265_amd.me_fullsearch(cur[:2], ref[:2], B, R, method="brute")
    same = torch.equal(brute.view(-1), mv[:2].view(-1)) if brute.dtype == mv.dtype else torch.equal(brute.reshape(-1).view(torch.int32), mv[:2].reshape(-1))
    print("blocks of %d: box %.4f  bounds %.4f  exact + list %.4f ms  (= every candidate on two frames: %s)" %
          (B, acc[0] / 8, acc[2] / 8, acc[3] / 8, same), flush=True)
