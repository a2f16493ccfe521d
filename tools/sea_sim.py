"""Offline model of the elimination search's batch policies (CPU only, numpy): for a sample of CTUs of the
bench's synthetic sequence, every candidate's SAD and every group's stored bound, then what each policy
for choosing the first batch would have evaluated.  A design aid, not a test."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x265_amd import synth  # noqa: E402

W, H, R, B = 1920, 1088, 64, 64
S = B // 4


def lb_bin(v):
    e = (np.asarray(v, np.float32).view(np.uint32) >> 21).astype(np.int64) - 507
    return np.maximum(e, 0)


def ctu_tables(cur, ref, cx, cy):
    dylo, dyhi = max(-R, -cy), min(R, H - B - cy)
    dxlo, dxhi = max(-R, -cx), min(R, W - B - cx)
    ndy, ndx = dyhi - dylo + 1, dxhi - dxlo + 1
    blk = cur[cy:cy + B, cx:cx + B].astype(np.int32)
    win = ref[cy + dylo:cy + dyhi + B, cx + dxlo:cx + dxhi + B].astype(np.int32)
    sad = np.empty((ndy, ndx), np.int64)
    v = np.lib.stride_tricks.sliding_window_view(win, (B, B))  # [ndy][ndx][B][B]
    for i in range(ndy):
        sad[i] = np.abs(v[i] - blk).sum(axis=(1, 2))
    # 16x16 sums
    ii = np.zeros((win.shape[0] + 1, win.shape[1] + 1), np.int64)
    ii[1:, 1:] = win.cumsum(0).cumsum(1)
    b16 = ii[S:, S:] - ii[:-S, S:] - ii[S:, :-S] + ii[:-S, :-S]   # b16[y][x] = sum of the SxS block at (x, y) of the window
    cs = blk.reshape(4, S, 4, S).sum(axis=(1, 3))
    lb = np.zeros((ndy, ndx), np.int64)
    for j in range(4):
        for k in range(4):
            lb += np.abs(b16[S * j:S * j + ndy, S * k:S * k + ndx] - cs[j, k])
    assert (lb <= sad).all()
    # groups: 4 window-aligned dx at one dy
    wx0 = (cx + dxlo) & ~3
    a = cx + dxlo - wx0
    ngx = (a + ndx + 3) >> 2
    lbp = np.full((ndy, 4 * ngx), 1 << 40, np.int64)
    sdp = np.full((ndy, 4 * ngx), 1 << 40, np.int64)
    lbp[:, a:a + ndx] = lb
    sdp[:, a:a + ndx] = sad
    gv = (lbp.reshape(ndy, ngx, 4).min(2) >> 4).ravel()     # stored bound of a group
    gs = sdp.reshape(ndy, ngx, 4).min(2).ravel()           # best exact SAD inside a group
    return gv, gs, ngx, ndy


def main():
    nsample = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seq = np.stack([synth.pad_plane(f, 64) for f in synth.make_sequence(1920, 1080, 4, search_range=R, seed=12345)])
    rng = np.random.default_rng(1)
    ncx, ncy = W // B, H // B
    rows = []
    for n in range(nsample):
        f = 1 + n % 3
        ctu = int(rng.integers(0, ncx * ncy))
        cx, cy = (ctu % ncx) * B, (ctu // ncx) * B
        gv, gs, ngx, ndy = ctu_tables(seq[f], seq[f - 1], cx, cy)
        ng = len(gv)
        best = gs.min()
        need = int((gv * 16 <= best).sum())
        order = np.argsort(lb_bin(gv), kind="stable")
        first = order[:256]
        best0 = gs[first].min()
        rest = np.ones(ng, bool)
        rest[first] = False
        n1 = int(((gv * 16 <= best0) & rest).sum())
        o = np.argsort(gv, kind="stable")
        ub1 = gs[o[0]]
        ub8 = gs[o[:8]].min()
        ub32 = gs[o[:32]].min()
        ub128 = gs[o[:128]].min()
        c1 = int((gv <= (ub1 >> 4)).sum())
        c8 = int((gv <= (ub8 >> 4)).sum())
        c32 = int((gv <= (ub32 >> 4)).sum())
        c128 = int((gv <= (ub128 >> 4)).sum())
        est = int((lb_bin(gv) <= lb_bin(ub1 >> 4)).sum())
        # eight disjoint interleaved subsets (what eight waves' own minima would be)
        sub = [np.arange(k, ng, 8) for k in range(8)]
        ubw = min(gs[s_[np.argmin(gv[s_])]] for s_ in sub)
        cw = int((gv <= (ubw >> 4)).sum())
        rows.append((ng, need, n1, est, c1, c8, cw, c32, c128, best, ub1, ub8, ub32))
    r = np.array(rows)
    names = ["ngroups", "need", "n1(now)", "est(now)", "cnt<=UB1", "cnt<=UB8", "cnt<=UBw8", "cnt<=UB32", "cnt<=UB128", "best", "ub1", "ub8", "ub32"]
    for i, nm in enumerate(names):
        q = np.percentile(r[:, i], [10, 25, 50, 75, 90, 99])
        print(f"{nm:12s} mean {r[:, i].mean():9.1f}  q10/25/50/75/90/99 {q.tolist()}")

    def rounds(n):
        return np.ceil(np.maximum(n, 1) / 128)

    light = r[:, 1] < 2048
    print("CTUs:", len(r), "light:", int(light.sum()))
    now = rounds(256) + np.where(r[:, 2] > 0, rounds(r[:, 2]), 0)
    print("rounds now (mean, light):", now[light].mean(), " second passes:", int((r[light, 2] > 0).sum()))
    for i, nm in ((4, "UB1"), (5, "UB8"), (6, "UBw8"), (7, "UB32"), (8, "UB128")):
        for cap in (256, 384, 512):
            c = r[:, i]
            one = c <= cap
            # one batch of everything under UB when it fits the cap; else 256 lowest + second pass as now
            rr = np.where(one, rounds(c), now)
            sp = np.where(one, 0, r[:, 2] > 0)
            print(f"  {nm} cap {cap}: single-batch CTUs {int(one[light].sum())}, mean rounds {rr[light].mean():.3f}, second passes {int(sp[light].sum())}")
    ideal = rounds(r[:, 1])
    print("rounds ideal (need only):", ideal[light].mean())


if __name__ == "__main__":
    main()
