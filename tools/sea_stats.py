"""Per-CTU statistics of the successive-elimination search (debug aid)."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265_amd
from x265_amd import synth, api
x265_amd.init(0)
W, H, R = (int(os.environ.get(k, d)) for k, d in (('SEA_W', 1920), ('SEA_H', 1088), ('SEA_R', 64)))
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seq = np.stack([synth.pad_plane(f, 64) for f in synth.make_sequence(W, H, NF + 1, search_range=R, seed=12345)])
fr = torch.from_numpy(seq).cuda()
nctu = (W // 64) * (H // 64)
stats = torch.zeros((NF, nctu, 16), dtype=torch.int32, device="cuda")
lib = api._lib()
lib.x265amd_debug_set_me_stats.argtypes = [C.c_void_p]
lib.x265amd_debug_set_me_stats(stats.data_ptr())
mv = x265_amd.me_fullsearch(fr[1:], fr[:-1], 64, R, method="sea")
torch.cuda.synchronize()
lib.x265amd_debug_set_me_stats(None)
s = stats.cpu().numpy().astype(np.int64).reshape(-1, 16)
nctu_all = s.shape[0]
names = ["n_pass0", "n_pass1", "cyc_stage", "cyc_pass0", "cyc_pass1"]
bnames = {10: "b_cyc_csum", 11: "b_cyc_bounds", 12: "b_cyc_scan", 13: "b_cyc_write"}
for i, n in enumerate(names):
    print(f"{n:14s} mean {s[:, i].mean():10.1f}  median {np.median(s[:, i]):10.1f}  max {s[:, i].max():10d}")
print(f"{'low_pass0':14s} mean {s[:, 15].mean():10.1f}  median {np.median(s[:, 15]):10.1f}  max {s[:, 15].max():10d}   (lower halves finished of the first batch)")
print(f"{'low_pass1':14s} mean {s[:, 14].mean():10.1f}  median {np.median(s[:, 14]):10.1f}  max {s[:, 14].max():10d}   (lower halves finished of the second pass)")
for i, n in bnames.items():
    print(f"{n:14s} mean {s[:, i].mean():10.1f}  median {np.median(s[:, i]):10.1f}  max {s[:, i].max():10d}")
interior = [i for i in range(nctu_all) if 2 <= (i % nctu) % (W // 64) <= W // 64 - 3 and 2 <= (i % nctu) // (W // 64) <= H // 64 - 3]
print("interior CTUs:", {n: float(s[interior, i].mean()) for i, n in enumerate(names)})
if s[:, 14].any():
    m = s[:, 1] > 0
    print("second pass, 16-wave shape: groups that outlive their upper half: mean %.1f of %.1f scheduled (%.0f %%)" %
          (s[m, 14].mean(), s[m, 1].mean(), 100.0 * s[m, 14].sum() / max(s[m, 1].sum(), 1)))
    print("  best SAD after the first batch vs final: ", "n/a")
print("CTUs with a second pass:", int((s[:, 1] > 0).sum()), "of", nctu_all,
      "| rounds per CTU (mean):", float((np.ceil(s[:, 0] / 128) + np.ceil(s[:, 1] / 128)).mean()))
print("pass-1 sizes, fine (1-4, 5-8, 9-16, 17-32, 33-64):",
      [int(((s[:, 1] >= a) & (s[:, 1] <= b)).sum()) for a, b in ((1, 4), (5, 8), (9, 16), (17, 32), (33, 64))],
      "| mean pass-1 cycles in those buckets:",
      [int(s[(s[:, 1] >= a) & (s[:, 1] <= b), 4].mean()) if ((s[:, 1] >= a) & (s[:, 1] <= b)).any() else 0 for a, b in ((1, 4), (5, 8), (9, 16), (17, 32), (33, 64))])
print("pass-1 sizes histogram (0, 1-64, 65-128, 129-512, >512):",
      [int(((s[:, 1] >= a) & (s[:, 1] <= b)).sum()) for a, b in ((0, 0), (1, 64), (65, 128), (129, 512), (513, 1 << 30))])

# per-CU timeline of the exact kernel: HW_ID bits cu_id[11:8] sh_id[12] se_id[15:13], XCC_ID in the high half
hw = s[:, 7]
cu = ((hw >> 8) & 0xff) | ((hw >> 16) << 8)
t0 = s[:, 5].astype(np.int64); t1 = s[:, 6].astype(np.int64)
t1 = np.where(t1 < t0, t1 + (1 << 32), t1)
spans = []; busy = []; conc = []; nwg = []
for c in sorted(set(cu.tolist())):
    m = cu == c
    ev = sorted([(a, 1) for a in t0[m]] + [(b, -1) for b in t1[m]])
    lvl = 0; last = ev[0][0]; b = 0; area = 0
    for t, d in ev:
        if lvl > 0: b += t - last
        area += lvl * (t - last)
        last = t; lvl += d
    sp = ev[-1][0] - ev[0][0]
    spans.append(sp); busy.append(b / sp); conc.append(area / max(b, 1)); nwg.append(int(m.sum()))
spans = np.array(spans)
print("CUs seen: %d | per-CU span (first WG start .. last WG end), cycles: mean %.0f max %.0f min %.0f" % (len(spans), spans.mean(), spans.max(), spans.min()))
print("WG residency mean %.0f | busy fraction of the CU's own span: mean %.3f min %.3f | concurrent WGs while busy: mean %.2f" % ((t1 - t0).mean(), np.mean(busy), np.min(busy), np.mean(conc)))
print("WGs per CU: min %d max %d mean %.1f" % (min(nwg), max(nwg), np.mean(nwg)))
xcc = np.array(sorted(set(cu.tolist()))) >> 8
for x in sorted(set(xcc.tolist())):
    mm = xcc == x
    print("  XCC %d: CUs %d, WGs %d, span mean %.0f max %.0f" % (x, mm.sum(), np.array(nwg)[mm].sum(), spans[mm].mean(), spans[mm].max()))

est = s[:, 8]; n1 = s[:, 1]
print("est >= 1024: %d CTUs; of the %d CTUs with n1 > 512, est >= 1024 for %d; est quantiles (10/50/90/99): %s" % (
    (est >= 1024).sum(), (n1 > 512).sum(), ((n1 > 512) & (est >= 1024)).sum(), np.percentile(est, [10, 50, 90, 99]).tolist()))
heavy = n1 > 512
print("heavy CTUs: start time relative to the CU's first start (cycles), quantiles 50/90/100:",
      np.percentile([t0[i] - t0[cu == cu[i]].min() for i in np.nonzero(heavy)[0]], [50, 90, 100]).tolist())
miss = np.nonzero((n1 > 512) & (est < 1024))[0]
print("missed heavy CTUs:", len(miss), "first (n1, est):", [(int(n1[i]), int(est[i])) for i in miss[:8]])
for lo, hi in ((0, 450), (450, 500), (500, 600), (600, 768), (768, 1024), (1024, 2048), (2048, 1 << 30)):
    m = (est >= lo) & (est < hi)
    if m.any():
        print("est [%d,%d): %d CTUs, n1 mean %.0f max %d, residency mean %.0f max %d" % (lo, hi, m.sum(), n1[m].mean(), n1[m].max(), (t1 - t0)[m].mean(), (t1 - t0)[m].max()))

# the CUs that finish last: what ran on them
order = np.argsort(-spans)[:4]
cus = sorted(set(cu.tolist()))
for oi in order:
    c = cus[oi]
    m = np.nonzero(cu == c)[0]
    base = t0[m].min()
    big = sorted([(int(t0[i] - base), int(t1[i] - t0[i]), int(n1[i]), int(est[i]), int(s[i, 9])) for i in m if t1[i] - t0[i] > 80000])
    print("CU %#x (XCC %d): span %d, %d CTUs; long ones (start, residency, n1, est, wg): %s" % (c, c >> 8, spans[oi], len(m), big[:12]))
# are the persistent workgroups spread evenly?  workgroup id -> CU
wg = s[:, 9]
per_cu_wgs = {}
for i in range(len(wg)):
    per_cu_wgs.setdefault(int(cu[i]), set()).add(int(wg[i]))
cnt = np.array([len(v) for v in per_cu_wgs.values()])
print("distinct persistent workgroups per CU: min %d max %d; CUs with 1: %d, 2: %d, 3+: %d" % (cnt.min(), cnt.max(), (cnt == 1).sum(), (cnt == 2).sum(), (cnt >= 3).sum()))
for x in sorted(set(xcc.tolist())):
    ws = set()
    for c, v in per_cu_wgs.items():
        if c >> 8 == x:
            ws |= v
    print("  XCC %d: %d persistent workgroups" % (x, len(ws)))
