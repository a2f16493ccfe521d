#!/usr/bin/env python3
"""The closed loop of bench.py --workload gop1080p (1080p x 16 segments, pictures back and forth) alone, for A/B runs.
python tools/gop_time.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import x265_amd  # noqa: E402
from x265_amd import ClosedLoopGop, synth  # noqa: E402

x265_amd.init(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
w, h, nseg, T, R = 1920, 1088, 16, 4, 64
segs = []
for s_ in range(nseg):
    y, u, v = synth.make_sequence(w, h, T, search_range=R, seed=12345 + s_, chroma=True)
    segs.append(np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1))
yuv = torch.from_numpy(np.stack(segs)).cuda()
pics = [yuv[:, t].contiguous() for t in list(range(1, T)) + list(range(T - 2, -1, -1))]
enc = ClosedLoopGop(w, h, nseg, search_range=R, qp=32)
for rnd in range(3):
    enc.set_reference(yuv[:, 0].contiguous())
    for k in range(20):
        enc.encode_p(pics[k % len(pics)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20, 20 + steps):
        enc.encode_p(pics[k % len(pics)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("round %d  %.4f ms per step  %.1f Gpix/s" % (rnd, dt * 1e3, 1920 * 1080 * nseg / dt / 1e9), flush=True)
