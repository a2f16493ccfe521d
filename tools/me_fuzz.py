#!/usr/bin/env python3
"""Random geometries through the three search formulations (every candidate, elimination, one candidate per lane)
and, on a sample of blocks, the CPU oracle: any disagreement is a bug.  Run on the GPU box:
    python tools/me_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import x265_amd as gpu  # noqa: E402
from x265_amd import api  # noqa: E402
from oracle import pyoracle  # noqa: E402


def content(rng, kind, f, r, h, w):
    if kind == "noise":
        cur = rng.integers(0, 256, (f, h, w), dtype=np.uint8)
        ref = rng.integers(0, 256, (f, r, h, w), dtype=np.uint8)
    elif kind == "flat":
        cur = np.full((f, h, w), int(rng.integers(0, 256)), np.uint8)
        ref = np.full((f, r, h, w), int(rng.integers(0, 256)), np.uint8)
    elif kind == "spots":  # flat with a few odd pixels: ties nearly everywhere, decided by raster order
        cur = np.full((f, h, w), 90, np.uint8)
        ref = np.full((f, r, h, w), 90, np.uint8)
        for a_ in (cur, ref):
            n = int(rng.integers(1, 40))
            flat = a_.reshape(-1)
            flat[rng.integers(0, flat.size, n)] = rng.integers(0, 256, n).astype(np.uint8)
    elif kind == "lowamp":  # two-bit noise: thousands of near ties, bounds close to the SADs' spread
        cur = rng.integers(0, 4, (f, h, w)).astype(np.uint8)
        ref = rng.integers(0, 4, (f, r, h, w)).astype(np.uint8)
    elif kind == "mixed":  # noise on one side of the picture, flat on the other, the references alike
        cur = rng.integers(0, 256, (f, h, w), dtype=np.uint8)
        ref = rng.integers(0, 256, (f, r, h, w), dtype=np.uint8)
        cut = int(rng.integers(1, w))
        cur[:, :, cut:] = 128
        ref[:, :, :, max(cut - 32, 0):] = 128
    elif kind == "repeat":  # a periodic pattern: many exact ties at different displacements
        px, py = int(rng.integers(2, 24)), int(rng.integers(2, 24))
        tile = rng.integers(0, 256, (py, px), dtype=np.uint8)
        big = np.tile(tile, (h // py + 2, w // px + 2))[:h, :w]
        cur = np.broadcast_to(big, (f, h, w)).copy()
        ref = np.broadcast_to(big, (f, r, h, w)).copy()
    else:  # a smooth picture + noise, the references shifted copies of it
        yy, xx = np.mgrid[0:h + 256, 0:w + 256]
        base = ((3 * xx + 2 * yy) >> 2) + rng.integers(0, 16, (h + 256, w + 256))
        cur = np.empty((f, h, w), np.uint8)
        ref = np.empty((f, r, h, w), np.uint8)
        for i in range(f):
            cur[i] = (base[128:128 + h, 128:128 + w] + rng.integers(0, 4, (h, w))).astype(np.uint8)
            for j in range(r):
                dy, dx = (int(v) for v in rng.integers(-100, 101, 2))
                ref[i, j] = (base[128 + dy:128 + dy + h, 128 + dx:128 + dx + w] + rng.integers(0, 4, (h, w))).astype(np.uint8)
    return cur, ref


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    gpu.init(0)
    orc = pyoracle.Oracle()
    bad = 0
    for case in range(cases):
        ctu = int(rng.choice([64, 64, 32]))
        w = ctu * int(rng.integers(1, 9))
        h = ctu * int(rng.integers(1, 7))
        rr = int(rng.choice([1, 3, 7, 16, 33, 64, 100, 128]))
        f, r = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        pad = int(rng.choice([0, 4, 64]))  # row stride = width + pad
        kind = str(rng.choice(["noise", "flat", "shift", "shift", "spots", "lowamp", "mixed", "repeat"]))
        cur, ref = content(rng, kind, f, r, h, w)
        # planes as views of wider buffers: the stride is not the width
        cbuf = torch.zeros((f, h, w + pad), dtype=torch.uint8, device="cuda")
        rbuf = torch.zeros((f, r, h, w + pad), dtype=torch.uint8, device="cuda")
        cbuf[:, :, :w] = torch.from_numpy(cur).cuda()
        rbuf[:, :, :, :w] = torch.from_numpy(ref).cuda()
        cv, rv = cbuf[:, :, :w], rbuf[:, :, :, :w]
        res = {}
        for m in ("brute", "sea", "basic"):
            try:
                res[m] = gpu.me_fullsearch(cv, rv, ctu, rr, method=m).cpu().numpy()
            except api.X265AmdError as e:  # a refusal (e.g. the window does not fit the LDS) is not a disagreement
                res[m] = None
                print(f"case {case}: {m} refused: {e}")
        ok = True
        for m in ("sea", "basic"):
            if res[m] is not None and res["brute"] is not None and not np.array_equal(res[m], res["brute"]):
                ok = False
                print(f"case {case}: {m} != brute")
        # the oracle on a few blocks of frame 0, reference 0
        nblk = (h // ctu) * (w // ctu)
        cnt = min(nblk, 3)
        step = max(nblk // cnt, 1)
        want = orc.me_fullsearch(cur[0], ref[0, 0], ctu, rr, first=0, step=step, count=cnt, nthreads=8)
        idx = np.arange(0, nblk, step)[:cnt]
        got = api.mv_to_numpy(torch.from_numpy(res["brute"]))[0, 0] if res["brute"] is not None else None
        if got is not None and not np.array_equal(got[idx], want[idx]):
            ok = False
            print(f"case {case}: brute != oracle on blocks {idx}")
        bad += not ok
        print(f"case {case}: {w}x{h} ctu {ctu} R {rr} frames {f} refs {r} stride +{pad} {kind}: {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
