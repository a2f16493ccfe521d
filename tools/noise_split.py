"""Per-kernel ms of the elimination search on content where nothing can be eliminated (debug aid)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265_amd
from x265_amd.pipeline import MeTqPipeline
x265_amd.init(0)
lib = x265_amd.api._lib()
g = np.random.default_rng(0)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, frames in (("noise", g.integers(0, 256, (nf + 1, 1088, 1920), dtype=np.uint8)),
                     ("flat", np.full((nf + 1, 1088, 1920), 77, np.uint8))):
    fr = torch.from_numpy(frames).cuda()
    # the two entry points alternately after a common warm-up (one after the other, the first pays for the clock ramp)
    pipes = {m: MeTqPipeline(fr, ctu=64, search_range=64, nrefs=1, method=m) for m in ("sea", "brute")}
    for _ in range(8):
        for m in pipes:
            pipes[m].motion_search()
    torch.cuda.synchronize()
    times = {m: [] for m in pipes}
    for _ in range(6):
        for m in pipes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pipes[m].motion_search()
            e1.record()
            torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1))
    lib.x265amd_debug_me_timing(1)
    buf = (ctypes.c_float * 4)()
    acc = np.zeros(4)
    for _ in range(5):
        pipes["sea"].motion_search()
        lib.x265amd_debug_me_last_ms(buf)
        acc += np.array(buf[:])
    lib.x265amd_debug_me_timing(0)
    print(f"{name} sea: {np.mean(times['sea']):.4f} ms  kernels (box, -, bounds, exact + list): " + str((acc / 5).round(4).tolist()))
    print(f"{name} brute: {np.mean(times['brute']):.4f} ms")
