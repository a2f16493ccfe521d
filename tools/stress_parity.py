"""One-off stress: the elimination search and the PU tree against the every-candidate kernels on random
geometries, ranges and content (synthetic motion, noise, flat, stripes, saturated).  Prints mismatches."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import x265_amd
from x265_amd import synth
x265_amd.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
for case in range(N):
    W, H = 64 * int(rng.integers(1, 9)), 64 * int(rng.integers(1, 7))
    R = int(rng.choice([0, 1, 2, 5, 8, 15, 16, 31, 47, 64, 70, 96]))
    F, NR = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    kind = rng.choice(["synth", "noise", "flat", "stripes", "sat", "mixed"])
    seq = synth.make_sequence(W, H, F + NR, search_range=max(2, min(R, 32)), seed=int(rng.integers(1 << 30))).copy()
    if kind == "noise":
        seq[:] = rng.integers(0, 256, seq.shape, dtype=np.uint8)
    elif kind == "flat":
        seq[:] = 128
    elif kind == "stripes":
        seq[:] = ((np.arange(W)[None, None, :] // 3 + np.arange(H)[None, :, None] // 5) % 2 * 255).astype(np.uint8)
    elif kind == "sat":
        seq[::2] = 255
        seq[1::2] = 0
    elif kind == "mixed":
        seq[:, : H // 2] = rng.integers(0, 256, (seq.shape[0], H // 2, W), dtype=np.uint8)
        seq[:, H // 2:, : W // 2] = 7
    fr = torch.from_numpy(seq).cuda()
    cur = fr[NR:]
    refs = torch.stack([fr[NR - 1 - r: NR - 1 - r + F] for r in range(NR)], dim=1)
    want = {b: x265_amd.me_fullsearch(cur, refs, b, R, method="brute") for b in (64, 32, 16)}
    sea = x265_amd.me_fullsearch(cur, refs, 64, R, method="auto")
    tree = x265_amd.me_fullsearch_pu(cur, refs, R) if R <= 100 else None
    ok = torch.equal(sea, want[64]) and (tree is None or all(torch.equal(tree[b], want[b]) for b in (64, 32, 16)))
    if not ok:
        bad += 1
        print("MISMATCH", case, W, H, R, F, NR, kind)
print("cases", N, "mismatches", bad)
