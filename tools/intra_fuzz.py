#!/usr/bin/env python3
"""Random picture sizes, CU sizes, quantisers, batch sizes and content through x265amd_intra_encode_frames against the
CPU oracle's restatement of the intra loop (reconstruction, levels, modes), and the host syntax writer against the
oracle's on the result.  Run on the GPU box:  python tools/intra_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import x265_amd as gpu  # noqa: E402
from x265_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    gpu.init(0)
    orc = pyoracle.Oracle()
    bad = 0
    for case in range(cases):
        cu = int(rng.choice([16, 32]))
        w = cu * int(rng.integers(1, 1 + 320 // cu))
        h = cu * int(rng.integers(1, 1 + 192 // cu))
        qp = int(rng.integers(0, 52))
        n = int(rng.integers(1, 6))
        fsz = w * h * 3 // 2
        frames = []
        for _ in range(n):
            kind = str(rng.choice(["noise", "flat", "probe", "blocks"]))
            if kind == "noise":
                frames.append(rng.integers(0, 256, fsz, dtype=np.uint8))
            elif kind == "flat":
                frames.append(np.full(fsz, int(rng.integers(0, 256)), np.uint8))
            elif kind == "probe":
                frames.append(synth.make_intra_probe(w, h, 1)[0])
            else:  # flat 8x8 blocks of random values: sharp edges, every direction useful somewhere
                y = np.kron(rng.integers(0, 256, (h // 8, w // 8)), np.ones((8, 8))).astype(np.uint8)
                c = np.kron(rng.integers(0, 256, (2, h // 16, w // 16)), np.ones((8, 8))).astype(np.uint8)
                frames.append(np.concatenate([y.ravel(), c.ravel()]))
        yuv = np.stack(frames)
        o = {k: t.cpu().numpy() for k, t in gpu.intra_encode_frames(torch.from_numpy(yuv).cuda(), w, h, cu, qp).items()}
        ok = True
        for i in range(n):
            orec, my, mc, lv = orc.intra_encode_frame(yuv[i], w, h, cu, qp, want_levels=True)
            same = (np.array_equal(o["rec"][i], orec) and np.array_equal(o["level"][i], lv) and np.array_equal(o["mode_y"][i], my) and
                    np.array_equal(o["mode_c"][i], mc))
            ok = ok and same
        bad += not ok
        print(f"case {case}: {w}x{h} cu {cu} qp {qp} batch {n}: {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
