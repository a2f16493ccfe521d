#!/usr/bin/env python3
"""Whole GOP segments on the GPU (x265_amd.ClosedLoopGop: the I picture through the intra path, the P pictures
searched against and predicted from the reconstruction before them) against the same chain on the CPU oracle, for
random picture sizes, ranges, quantisers and segment counts.  Run on the GPU box:  python tools/gop_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import x265_amd as gpu  # noqa: E402
from x265_amd import synth, ClosedLoopGop  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    gpu.init(0)
    orc = pyoracle.Oracle()
    bad = 0
    for case in range(cases):
        W, H = 64 * int(rng.integers(1, 6)), 64 * int(rng.integers(1, 4))
        R = int(rng.choice([3, 16, 33, 64]))
        QP = int(rng.integers(8, 46))
        S, T = int(rng.integers(1, 4)), 3
        CU = int(rng.choice([16, 32]))
        ysz, csz = W * H, (W // 2) * (H // 2)
        segs = []
        for s in range(S):
            y, u, v = synth.make_sequence(W, H, T, search_range=min(R, 32), seed=int(rng.integers(1, 1000)), chroma=True)
            segs.append(np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1))
        yuv = np.stack(segs)
        enc = ClosedLoopGop(W, H, S, search_range=R, qp=QP)
        rec0 = enc.encode_i(torch.from_numpy(yuv[:, 0]).cuda(), cu=CU).cpu().numpy()
        ok = True
        ref_cpu = []
        for s in range(S):
            orec, my, mc, lv = orc.intra_encode_frame(yuv[s, 0], W, H, CU, QP, want_levels=True)
            ok = ok and np.array_equal(rec0[s], orec) and np.array_equal(enc.level.cpu().numpy()[s], lv)
            ref_cpu.append(orec)
        qpc = orc.chroma_qp(QP)
        for t in range(1, T):
            rec_gpu = enc.encode_p(torch.from_numpy(yuv[:, t]).cuda()).cpu().numpy()
            gmv = gpu.mv_to_numpy(enc.mv)[:, 0]
            glevel = enc.level.cpu().numpy()
            for s in range(S):
                cur, ref = yuv[s, t], ref_cpu[s]
                cy, ry = cur[:ysz].reshape(H, W), ref[:ysz].reshape(H, W)
                omv = orc.me_fullsearch(cy, ry, 64, R, nthreads=8)
                ok = ok and np.array_equal(gmv[s], omv)
                ly, _ = orc.tq_forward(cy, ry, 32, QP, 1, mv=omv, mv_ctu=64)
                rec = [orc.tq_inverse(ly, ry, 32, QP, 1, mv=omv, mv_ctu=64).reshape(-1)]
                lev = [ly.reshape(-1)]
                cmv = omv.copy()
                cmv["mvx"] >>= 1
                cmv["mvy"] >>= 1
                for k in range(2):
                    a, b = ysz + k * csz, ysz + (k + 1) * csz
                    cc, rc = cur[a:b].reshape(H // 2, W // 2), ref[a:b].reshape(H // 2, W // 2)
                    lc, _ = orc.tq_forward(cc, rc, 16, qpc, 1, mv=cmv, mv_ctu=32)
                    rec.append(orc.tq_inverse(lc, rc, 16, qpc, 1, mv=cmv, mv_ctu=32).reshape(-1))
                    lev.append(lc.reshape(-1))
                ref_cpu[s] = np.concatenate(rec)
                ok = ok and np.array_equal(glevel[s], np.concatenate(lev)) and np.array_equal(rec_gpu[s], ref_cpu[s])
        bad += not ok
        print(f"case {case}: {W}x{H} R {R} QP {QP} segments {S} intra CU {CU}: {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
