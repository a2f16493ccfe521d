#!/usr/bin/env python3
"""usage: python tools/tq_trace.py [tu] [frames] [units per workgroup, 0 = even split]
Where the time of one launch of the matrix-core TQ kernel goes, from the per-wave records of
x265amd_debug_tq_trace (include/x265_amd_debug.h): launch ramp, spread of the wave durations, tail, shader
clock, workgroups per CU.  Run on the GPU box:  python tools/tq_trace.py [tu] [frames]"""
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import x265_amd as gpu  # noqa: E402
from x265_amd import api  # noqa: E402


def main():
    tu = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    gpu.init(0)
    api._lib().x265amd_debug_tq_units_per_wg(per)
    g = torch.Generator(device="cpu").manual_seed(1)
    src = torch.randint(0, 256, (nf, 1088, 1920), dtype=torch.uint8, generator=g).cuda()
    pred = torch.randint(0, 256, (nf, 1088, 1920), dtype=torch.uint8, generator=g).cuda()
    out = None
    level, sums, rec = gpu.tq_roundtrip(src, pred, tu, 32, gpu.SLICE_P)
    out = (level, sums, rec)
    for _ in range(5):
        gpu.tq_roundtrip(src, pred, tu, 32, gpu.SLICE_P, out=out)
    torch.cuda.synchronize()
    # back-to-back launches, as the bench times them
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
        gpu.tq_roundtrip(src, pred, tu, 32, gpu.SLICE_P, out=out)
    ev1.record()
    torch.cuda.synchronize()
    print(f"units per workgroup {per}: {ev0.elapsed_time(ev1) * 20:.1f} us per launch over 50 launches")
    nw = 4 * nf * 34 * 15
    buf = torch.zeros(nw * 4, dtype=torch.int64, device="cuda")
    api._lib().x265amd_debug_tq_trace(buf.data_ptr(), nw)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    gpu.tq_roundtrip(src, pred, tu, 32, gpu.SLICE_P, out=out)
    ev1.record()
    torch.cuda.synchronize()
    api._lib().x265amd_debug_tq_trace(None, 0)
    r = buf.cpu().numpy().reshape(-1, 4)
    r = r[r[:, 1] != 0]
    t0, t1, clk, hw = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    base = t0.min()
    s = (t0 - base) * 0.01   # us (100 MHz)
    e = (t1 - base) * 0.01
    dur = e - s
    print(f"TU {tu}, {nf} frames: event time {ev0.elapsed_time(ev1) * 1000:.1f} us; waves traced {len(r)}")
    q = lambda a: " ".join(f"{np.percentile(a, p):7.2f}" for p in (0, 10, 50, 90, 99, 100))
    print("percentiles           0      10      50      90      99     100")
    print("wave start  (us) ", q(s))
    print("wave end    (us) ", q(e))
    print("wave length (us) ", q(dur))
    print("shader clock (GHz)", q(clk / (dur * 1000.0 + 1e-9)))
    xcc = (hw >> 32) & 0xf
    hwid = hw & 0xffffffff
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    simd = (hwid >> 4) & 0x3
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    cus, cnt = np.unique(key, return_counts=True)
    print(f"CUs used {len(cus)}; waves per CU: min {cnt.min()} max {cnt.max()} histogram {np.bincount(cnt)[1:].tolist()} (index = count - 1)")
    skey = key * 4 + simd
    _, scnt = np.unique(skey, return_counts=True)
    print(f"waves per SIMD: histogram {np.bincount(scnt).tolist()}")
    # mean concurrency over the launch
    span = e.max()
    print(f"span of wave activity {span:.2f} us; mean waves resident {dur.sum() / span:.0f} of {len(r)}")
    # start time by XCC
    for x in np.unique(xcc):
        m = xcc == x
        print(f"  XCC {x}: waves {m.sum():5d} start median {np.median(s[m]):6.2f} end median {np.median(e[m]):6.2f} end max {e[m].max():6.2f}")


if __name__ == "__main__":
    main()
