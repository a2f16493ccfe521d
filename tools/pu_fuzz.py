#!/usr/bin/env python3
"""Random geometries through the per-PU searches (x265amd_me_fullsearch_pu, _pu_all) against the every-candidate
search at the PU's size and, for the rectangles and 8x8, the CPU oracle on a sample of blocks.
Run on the GPU box:  python tools/pu_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x265_amd as gpu  # noqa: E402
from x265_amd import api  # noqa: E402
from oracle import pyoracle  # noqa: E402
from me_fuzz import content  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    gpu.init(0)
    orc = pyoracle.Oracle()
    bad = 0
    for case in range(cases):
        w, h = 64 * int(rng.integers(1, 5)), 64 * int(rng.integers(1, 4))
        rr = int(rng.choice([1, 5, 16, 40, 64, 100, 128]))
        f, r = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        kind = str(rng.choice(["noise", "flat", "shift", "shift"]))
        cur, ref = content(rng, kind, f, r, h, w)
        cv, rv = torch.from_numpy(cur).cuda(), torch.from_numpy(ref).cuda()
        ok = True
        try:
            tree = gpu.me_fullsearch_pu(cv, rv, rr)
            allp = gpu.me_fullsearch_pu_all(cv, rv, rr)
        except api.X265AmdError as e:
            print(f"case {case}: {w}x{h} R {rr}: refused: {e}")
            continue
        for b in (64, 32, 16):
            want = gpu.me_fullsearch(cv, rv, b, rr, method="brute").cpu().numpy()
            if not np.array_equal(tree[b].cpu().numpy(), want):
                ok = False
                print(f"case {case}: tree {b} != brute")
            if not np.array_equal(allp[(b, b)].cpu().numpy().reshape(want.shape), want):
                ok = False
                print(f"case {case}: pu_all {b}x{b} != brute")
        for (bw, bh) in ((64, 32), (32, 64), (32, 16), (16, 32), (16, 8), (8, 16), (8, 8)):
            nb = (h // bh) * (w // bw)
            cnt = min(nb, 4)
            step = max(nb // cnt, 1)
            want = orc.me_fullsearch_wh(cur[0], ref[0, 0], bw, bh, rr, first=0, step=step, count=cnt, nthreads=8).ravel()
            got = api.mv_to_numpy(allp[(bw, bh)])[0, 0].ravel()
            idx = np.arange(0, nb, step)[:cnt]
            if not np.array_equal(got[idx], want[idx]):
                ok = False
                print(f"case {case}: pu_all {bw}x{bh} != oracle on blocks {idx}")
        bad += not ok
        print(f"case {case}: {w}x{h} R {rr} frames {f} refs {r} {kind}: {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
