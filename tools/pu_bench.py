"""PU-size searches at 1080p, +-64: the per-size every-candidate kernel, the successive-elimination CTU search,
and the one-pass tree (x265amd_me_fullsearch_pu) that gives all three sizes at once."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import x265_amd
from x265_amd import synth
x265_amd.init(0)
W, H, R, NF = 1920, 1088, 64, 8
seq = np.stack([synth.pad_plane(f, 64) for f in synth.make_sequence(1920, 1080, NF + 1, search_range=R, seed=12345)])
fr = torch.from_numpy(seq).cuda()


def timed(fn):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return r, dt


res = {}
for ctu in (64, 32, 16):
    res[ctu], dt = timed(lambda: x265_amd.me_fullsearch(fr[1:], fr[:-1], ctu, R, method="brute"))
    print("PU %2d  every candidate      %6.2f ms  %6.0f Mpix/s" % (ctu, dt * 1e3, 1920 * 1080 * NF / dt / 1e6))
_, dt = timed(lambda: x265_amd.me_fullsearch(fr[1:], fr[:-1], 64, R, method="sea"))
print("PU 64  successive elimination %6.2f ms  %6.0f Mpix/s" % (dt * 1e3, 1920 * 1080 * NF / dt / 1e6))
ws32 = torch.empty(x265_amd.me_workspace_bytes(1920, 1088, R, NF, 1, 32), dtype=torch.uint8, device="cuda")
sea32, dt = timed(lambda: x265_amd.me_fullsearch(fr[1:], fr[:-1], 32, R, method="sea", workspace=ws32))
print("PU 32  successive elimination %6.2f ms  %6.0f Mpix/s   identical to every candidate: %s" %
      (dt * 1e3, 1920 * 1080 * NF / dt / 1e6, torch.equal(sea32, res[32])))
tree, dt = timed(lambda: x265_amd.me_fullsearch_pu(fr[1:], fr[:-1], R))
print("PU 64 + 32 + 16, one pass     %6.2f ms  %6.0f Mpix/s" % (dt * 1e3, 1920 * 1080 * NF / dt / 1e6))
t2, dt = timed(lambda: x265_amd.me_fullsearch_pu(fr[1:], fr[:-1], R, sizes=(32, 16)))
print("PU 32 + 16, one pass          %6.2f ms  %6.0f Mpix/s" % (dt * 1e3, 1920 * 1080 * NF / dt / 1e6))
print("identical to the per-size kernels:", all(torch.equal(tree[b], res[b]) for b in (64, 32, 16)))
allpu, dt = timed(lambda: x265_amd.me_fullsearch_pu_all(fr[1:], fr[:-1], R))
print("all ten PU shapes (169 PUs per CTU), one pass %6.2f ms  %6.0f Mpix/s" % (dt * 1e3, 1920 * 1080 * NF / dt / 1e6))
print("squares identical to the per-size kernels:",
      all(torch.equal(allpu[(b, b)].reshape(res[b].shape), res[b]) for b in (64, 32, 16)))
