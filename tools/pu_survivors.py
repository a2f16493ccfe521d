#!/usr/bin/env python3
"""What displacement elimination could save the all-shapes PU search (x265amd_me_fullsearch_pu_all): for sampled CTUs,
the share of displacement groups (one dy, four dx: the kernel's task) that SOME PU of the CTU still needs when every
8x8 block b carries the threshold t_b = max over the PUs p that contain it of SAD_b(best vector of p) -- the best
thresholds a per-block test can have (they come from the final answer) -- and a displacement is dropped only if
LB_b(d) > t_b for every block.  LB_b from one 8x8 box sum (lb8) or from four 4x4 box sums (lb4).
CPU only (numpy):  python tools/pu_survivors.py [ctus]  > profiles/r3_pu_survivors.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from x265_amd import synth  # noqa: E402

R = 64
SPAN = 2 * R + 1
PUS = [(y, x, h, w) for (h, w) in ((8, 8), (8, 4), (4, 8), (4, 4), (4, 2), (2, 4), (2, 2), (2, 1), (1, 2), (1, 1))
       for y in range(0, 8, h) for x in range(0, 8, w)]  # in units of 8x8 blocks: 169 PUs


def ctu_shares(cur, ref, cx, cy):
    c = cur[cy:cy + 64, cx:cx + 64]
    sad8 = np.zeros((SPAN, SPAN, 8, 8), np.int32)
    lb8 = np.zeros_like(sad8)
    lb4 = np.zeros_like(sad8)
    cs8 = c.reshape(8, 8, 8, 8).sum(axis=(1, 3))
    cs4 = c.reshape(16, 4, 16, 4).sum(axis=(1, 3))
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            r = ref[cy + dy:cy + dy + 64, cx + dx:cx + dx + 64]
            sad8[dy + R, dx + R] = np.abs(c - r).reshape(8, 8, 8, 8).sum(axis=(1, 3))
            lb8[dy + R, dx + R] = np.abs(cs8 - r.reshape(8, 8, 8, 8).sum(axis=(1, 3)))
            lb4[dy + R, dx + R] = np.abs(cs4 - r.reshape(16, 4, 16, 4).sum(axis=(1, 3))).reshape(8, 2, 8, 2).sum(axis=(1, 3))
    t = np.zeros((8, 8), np.int64)
    for (y, x, h, w) in PUS:
        s = sad8[:, :, y:y + h, x:x + w].sum(axis=(2, 3))
        dyb, dxb = divmod(int(np.argmin(s)), SPAN)  # first minimum in raster order
        t[y:y + h, x:x + w] = np.maximum(t[y:y + h, x:x + w], sad8[dyb, dxb, y:y + h, x:x + w])
    out = []
    for lb in (lb8, lb4):
        surv = (lb <= t[None, None]).any(axis=(2, 3))
        g = np.zeros((SPAN, (SPAN + 3) // 4), bool)
        for q in range(4):
            sl = surv[:, q::4]
            g[:, :sl.shape[1]] |= sl
        out.append(float(g.mean()))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    for name, seq in (("make_sequence (the benchmark's content)", synth.make_sequence(1920, 1080, 2, search_range=R, seed=12345)),
                      ("make_textured_sequence (bench.py mid_case)", synth.make_textured_sequence(1920, 1080, 2, search_range=R))):
        seq = np.stack([synth.pad_plane(f, 64) for f in seq])
        cur, ref = seq[1].astype(np.int32), seq[0].astype(np.int32)
        rng = np.random.default_rng(3)
        res = []
        print(name)
        for _ in range(n):
            cx, cy = 64 * int(rng.integers(1, 28)), 64 * int(rng.integers(1, 15))  # interior CTUs
            res.append(ctu_shares(cur, ref, cx, cy))
            print("  CTU (%4d, %4d): groups kept with 8x8 sums %.3f, with 4x4 sums %.3f" % (cx, cy, *res[-1]), flush=True)
        m = np.mean(res, axis=0)
        print("  mean over %d CTUs: 8x8 sums %.3f, 4x4 sums %.3f" % (n, m[0], m[1]))


if __name__ == "__main__":
    main()
