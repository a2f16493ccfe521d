"""Per-step time of the closed-loop workload (bench.py --workload gop1080p): which pictures are the slow ones.
python tools/gop_steps.py [T] [order]   order = wrap (bench.py's 1..T-1, 1..T-1, ..) or pingpong (1..T-1, T-2..0, 1..)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265_amd
from x265_amd import synth, ClosedLoopGop
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
order = sys.argv[2] if len(sys.argv) > 2 else "wrap"
w, h, nseg, rng, qp = 1920, 1088, 16, 64, 32
x265_amd.init(0)
segs = []
for s_ in range(nseg):
    y, u, v = synth.make_sequence(w, h, T, search_range=rng, seed=12345 + s_, chroma=True)
    segs.append(np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1))
yuv = torch.from_numpy(np.stack(segs)).cuda()
enc = ClosedLoopGop(w, h, nseg, search_range=rng, qp=qp)
enc.set_reference(yuv[:, 0].contiguous())
if order == "wrap":
    seq = [1 + k % (T - 1) for k in range(4 * (T - 1))]
else:
    lap = list(range(1, T)) + list(range(T - 2, -1, -1))
    seq = (lap * 4)[:4 * (T - 1)]
pics = [yuv[:, t].contiguous() for t in range(T)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)]
ev[0].record()
for i, t in enumerate(seq):
    enc.encode_p(pics[t])
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(seq))]
print(order, "T =", T)
print(" ".join(f"{t}:{m:.3f}" for t, m in zip(seq, ms)))
print("mean over all but the first lap: %.4f ms -> %.1f Gpix/s" % (np.mean(ms[T - 1:]), 1920 * 1080 * nseg / np.mean(ms[T - 1:]) / 1e6))
