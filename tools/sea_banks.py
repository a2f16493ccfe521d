"""LDS bank-conflict model of the exact kernel's window reads for different window pitches (debug aid).
Reads the first-batch schedule (list0) back from the workspace of a real run."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265_amd
from x265_amd import synth, api
x265_amd.init(0)
W, H, R, NF = 1920, 1088, 64, 2
seq = np.stack([synth.pad_plane(f, 64) for f in synth.make_sequence(1920, 1080, NF + 1, search_range=R, seed=12345)])
fr = torch.from_numpy(seq).cuda()
nctu = (W // 64) * (H // 64)
ws = torch.zeros(api.me_workspace_bytes(W, H, R, NF, 1), dtype=torch.uint8, device="cuda")
mv = x265_amd.me_fullsearch(fr[1:], fr[:-1], 64, R, method="sea", workspace=ws)
torch.cuda.synchronize()
a256 = lambda x: (x + 255) & ~255
gl_pitch = (33 * 129 + 7) & ~7
off_gl = a256(6 * W * H * NF)
off_l0 = off_gl + a256(2 * gl_pitch * nctu * NF)
l0 = ws[off_l0:off_l0 + 2 * gl_pitch * nctu * NF].cpu().numpy().view(np.uint16).reshape(NF * nctu, gl_pitch)
ncx = W // 64
for pitch in (0, 49, 51, 52):  # 0 = the kernel's own pitch for that CTU
    tot = 0; n = 0
    for c in range(0, NF * nctu, 7):
        cin = c % nctu; cx = (cin % ncx) * 64; cy = (cin // ncx) * 64
        dxlo = max(-R, -cx); dxhi = min(R, W - 64 - cx)
        wx0 = (cx + dxlo) & ~3
        ngx = (cx + dxlo - wx0 + (dxhi - dxlo + 1) + 3) >> 2
        ent = l0[c, :256].astype(np.int64)
        dyi, gx = ent // ngx, ent % ngx
        bank = (dyi * (pitch if pitch else ((ngx + 17) | 1)) + gx) % 32
        for h in range(0, 256, 32):
            tot += np.bincount(bank[h:h + 32], minlength=32).max(); n += 1
    print("pitch %3d: mean conflict degree per 32-lane access %.2f" % (pitch, tot / n))
