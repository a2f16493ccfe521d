"""Per-CTU statistics of the elimination search inside the closed loop (bench.py --workload gop1080p): the reference is a
RECONSTRUCTION.  python tools/sea_stats_gop.py [pictures before the measured one]"""
import ctypes as C
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265_amd
from x265_amd import synth, api, ClosedLoopGop
x265_amd.init(0)
w, h, nseg, rng, qp, T = 1920, 1088, 16, 64, 32, 4
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 5
segs = []
for s_ in range(nseg):
    y, u, v = synth.make_sequence(w, h, T, search_range=rng, seed=12345 + s_, chroma=True)
    segs.append(np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1))
yuv = torch.from_numpy(np.stack(segs)).cuda()
enc = ClosedLoopGop(w, h, nseg, search_range=rng, qp=qp)
enc.set_reference(yuv[:, 0].contiguous())
order = (list(range(1, T)) + list(range(T - 2, -1, -1))) * 8
pics = [yuv[:, t].contiguous() for t in range(T)]
for t in order[:warm]:
    enc.encode_p(pics[t])
nctu = (w // 64) * (h // 64)
stats = torch.zeros((nseg, nctu, 16), dtype=torch.int32, device="cuda")
lib = api._lib()
lib.x265amd_debug_set_me_stats.argtypes = [C.c_void_p]
lib.x265amd_debug_set_me_stats(stats.data_ptr())
enc.encode_p(pics[order[warm]])
torch.cuda.synchronize()
lib.x265amd_debug_set_me_stats(None)
s = stats.cpu().numpy().astype(np.int64).reshape(-1, 16)
names = ["n_pass0", "n_pass1", "cyc_stage", "cyc_pass0", "cyc_pass1"]
for i, n in enumerate(names):
    print(f"{n:14s} mean {s[:, i].mean():10.1f}  median {np.median(s[:, i]):10.1f}  p90 {np.percentile(s[:, i], 90):10.0f}  max {s[:, i].max():10d}")
print(f"{'low_pass0':14s} mean {s[:, 15].mean():10.1f}  median {np.median(s[:, 15]):10.1f}  max {s[:, 15].max():10d}")
print(f"{'low_pass1':14s} mean {s[:, 14].mean():10.1f}  median {np.median(s[:, 14]):10.1f}  max {s[:, 14].max():10d}")
print(f"{'est':14s} quantiles 10/50/90/99: {np.percentile(s[:, 8], [10, 50, 90, 99])}")
print("pass-1 sizes histogram (0, 1-64, 65-128, 129-512, >512):",
      [int(((s[:, 1] >= a) & (s[:, 1] <= b)).sum()) for a, b in ((0, 0), (1, 64), (65, 128), (129, 512), (513, 1 << 30))])
tot = s[:, 2] + s[:, 3] + s[:, 4]
print("cycles per CTU (stage + pass0 + pass1): mean %.0f; share of CTUs with n1 > 512 in the total: %.2f" %
      (tot.mean(), tot[s[:, 1] > 512].sum() / tot.sum()))
mvs = enc.mv.cpu().numpy().reshape(-1, 2)
sad = (mvs[:, 1].astype(np.int64)) if mvs.shape[1] == 2 else None
