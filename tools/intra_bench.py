#!/usr/bin/env python3
"""One intra frame on the GPU (x265amd_intra_encode_frame): time per frame launched eagerly and replayed from a
captured graph (the frame is 3 launches per anti-diagonal of CUs: launch-bound), beside the CPU reference /
restatement on this box's host.  Run on the GPU box:  python tools/intra_bench.py [--two-launches]
(--two-launches: the transform stage of a step as a launch of its own, x265amd_debug_intra_fused(0), for comparison)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import x265_amd as gpu  # noqa: E402
from x265_amd import api, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402


def planes(yuv, w, h):
    return (yuv[:w * h].reshape(h, w), yuv[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), yuv[w * h * 5 // 4:].reshape(h // 2, w // 2))


def main():
    gpu.init(0)
    orc = pyoracle.Oracle()
    if "--two-launches" in sys.argv[1:]:
        api._lib().x265amd_debug_intra_fused(0)
        print("prediction kernel and transform stage as two launches per step")
    for a in sys.argv[1:]:
        if a.startswith("--threads="):
            api._lib().x265amd_debug_intra_threads(int(a.split("=")[1]))
            print(f"prediction kernel forced to {a.split('=')[1]} threads per workgroup")
    for (w, h, cu, qp) in ((416, 240, 16, 32), (1920, 1088, 32, 32), (3840, 2176, 32, 32)):
        yuv = synth.make_intra_probe(w, h, 1)[0]
        chroma = torch.from_numpy(np.ascontiguousarray(yuv[w * h:]).reshape(2, h // 2, w // 2)).cuda()
        y = torch.from_numpy(np.ascontiguousarray(yuv[:w * h]).reshape(h, w)).cuda()
        u, v = chroma[0], chroma[1]
        out = api.intra_frame_buffers(h, w, cu, y.device)
        for _ in range(2):
            gpu.intra_encode_frame(y, u, v, cu, qp, out=out)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            gpu.intra_encode_frame(y, u, v, cu, qp, out=out)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n
        ref_rec = torch.cat([out["rec_y"].flatten(), out["rec_u"].flatten(), out["rec_v"].flatten()]).cpu().numpy()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            gpu.intra_encode_frame(y, u, v, cu, qp, out=out)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            gpu.intra_encode_frame(y, u, v, cu, qp, out=out)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n
        rec = torch.cat([out["rec_y"].flatten(), out["rec_u"].flatten(), out["rec_v"].flatten()]).cpu().numpy()
        t0 = time.perf_counter()
        orec, my, mc, _ = orc.intra_encode_frame(yuv, w, h, cu, qp)
        cpu = time.perf_counter() - t0
        nw = w // cu + 2 * (h // cu - 1)
        print(f"{w}x{h} CU{cu}: {nw} anti-diagonals; eager {eager * 1e3:7.2f} ms/frame, graph replay {graph * 1e3:7.2f} ms/frame "
              f"({w * h / graph / 1e6:7.1f} Mpix/s); CPU restatement, one core {cpu * 1e3:8.1f} ms/frame; "
              f"identical to it: {np.array_equal(rec, orec) and np.array_equal(ref_rec, orec)}")
        # a batch of independent frames per pass over the anti-diagonals (x265amd_intra_encode_frames)
        for nb in (4, 8, 16, 32, 64):
            if nb * w * h * 3 // 2 * 5 > 6e9:
                continue
            seq = synth.make_intra_probe(w, h, min(nb, 8))
            batch = torch.from_numpy(np.ascontiguousarray(np.resize(seq, (nb, w * h * 3 // 2)))).cuda()
            o = gpu.intra_encode_frames(batch, w, h, cu, qp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                gpu.intra_encode_frames(batch, w, h, cu, qp, out=o)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            same = np.array_equal(o["rec"][0].cpu().numpy(), orec)
            print(f"    batch of {nb:2d} frames: {dt * 1e3:7.2f} ms per batch = {dt / nb * 1e3:6.3f} ms/frame "
                  f"({nb * w * h / dt / 1e6:8.1f} Mpix/s); frame 0 identical: {same}")


if __name__ == "__main__":
    main()
