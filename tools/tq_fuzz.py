#!/usr/bin/env python3
"""Random plane geometries, TU sizes, quantisers, roundings, strides and per-TU modes through the transform / quant
entry points against the CPU oracle: any difference is a bug.  Run on the GPU box:  python tools/tq_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import x265_amd as gpu  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    gpu.init(0)
    orc = pyoracle.Oracle()
    bad = 0
    for case in range(cases):
        tu = int(rng.choice([4, 8, 16, 32]))
        w = tu * int(rng.integers(1, 1 + 416 // tu))
        h = tu * int(rng.integers(1, 1 + 256 // tu))
        f = int(rng.integers(1, 4))
        qp, sl = int(rng.integers(0, 52)), int(rng.integers(0, 3))
        pad_s, pad_p = int(rng.choice([0, 1, 16, 100])), int(rng.choice([0, 3, 16, 64]))
        kind = str(rng.choice(["noise", "extreme", "near"]))
        if kind == "noise":
            src = rng.integers(0, 256, (f, h, w), dtype=np.uint8)
            pred = rng.integers(0, 256, (f, h, w), dtype=np.uint8)
        elif kind == "extreme":
            src = rng.choice(np.array([0, 255], np.uint8), (f, h, w))
            pred = rng.choice(np.array([0, 255], np.uint8), (f, h, w))
        else:
            pred = rng.integers(0, 256, (f, h, w), dtype=np.uint8)
            src = np.clip(pred.astype(np.int32) + rng.integers(-6, 7, (f, h, w)), 0, 255).astype(np.uint8)
        modes = None
        if tu == 4 and rng.integers(0, 2):
            modes = rng.choice(np.array([0, 1, 2, 10, 11, 25, 26, 34, 255], np.uint8), (f, (h // 4) * (w // 4)))
        # motion-compensated prediction: one vector per mv_ctu x mv_ctu block, every block stays inside the plane
        mv = mvt = None
        mv_ctu = int(rng.choice([16, 32, 64]))
        if tu <= mv_ctu and w % mv_ctu == 0 and h % mv_ctu == 0 and modes is None and rng.integers(0, 2):
            ncx, ncy = w // mv_ctu, h // mv_ctu
            mv = np.zeros((f, ncy * ncx), pyoracle.MV_DTYPE)
            for i in range(f):
                for c in range(ncy * ncx):
                    cx, cy = (c % ncx) * mv_ctu, (c // ncx) * mv_ctu
                    mv[i, c]["mvx"] = int(rng.integers(-cx, w - mv_ctu - cx + 1))
                    mv[i, c]["mvy"] = int(rng.integers(-cy, h - mv_ctu - cy + 1))
            mvt = torch.from_numpy(mv.view(np.int32).reshape(f, ncy * ncx, 2)).cuda()
        sbuf = torch.zeros((f, h, w + pad_s), dtype=torch.uint8, device="cuda")
        pbuf = torch.zeros((f, h, w + pad_p), dtype=torch.uint8, device="cuda")
        sbuf[:, :, :w] = torch.from_numpy(src).cuda()
        pbuf[:, :, :w] = torch.from_numpy(pred).cuda()
        sv, pv = sbuf[:, :, :w], pbuf[:, :, :w]
        md = torch.from_numpy(modes).cuda() if modes is not None else None
        level, sums, rec = gpu.tq_roundtrip(sv, pv, tu, qp, sl, modes=md, mv=mvt, mv_ctu=mv_ctu)
        lf, sf = gpu.tq_forward(sv, pv, tu, qp, sl, modes=md, mv=mvt, mv_ctu=mv_ctu)
        ri = gpu.tq_inverse(lf, pv, tu, qp, sl, modes=md, mv=mvt, mv_ctu=mv_ctu)
        level, sums, rec, lf, sf, ri = (t.cpu().numpy() for t in (level, sums, rec, lf, sf, ri))
        ok = np.array_equal(level, lf) and np.array_equal(sums, sf) and np.array_equal(rec, ri)
        for i in range(f):
            m = modes[i] if modes is not None else None
            mvi = np.ascontiguousarray(mv[i]) if mv is not None else None
            ol, os_ = orc.tq_forward(src[i], pred[i], tu, qp, sl, modes=m, mv=mvi, mv_ctu=mv_ctu)
            orec = orc.tq_inverse(ol, pred[i], tu, qp, sl, modes=m, mv=mvi, mv_ctu=mv_ctu)
            ok = ok and np.array_equal(level[i], ol) and np.array_equal(sums[i].ravel().astype(np.uint32), os_) and np.array_equal(rec[i], orec)
        bad += not ok
        print(f"case {case}: {w}x{h} x{f} tu {tu} qp {qp} slice {sl} strides +{pad_s}/+{pad_p} {kind}"
              f"{' modes' if modes is not None else ''}{f' vectors per {mv_ctu}' if mv is not None else ''}: {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
