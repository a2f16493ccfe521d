"""Per-kernel ms of the elimination search on the bench's 1080p sequence (16 frames): HIP events of the library,
averaged over 20 calls after 10 warm-up calls.  For A/B runs (tools/ab.sh)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265_amd
from x265_amd import synth
from x265_amd.pipeline import MeTqPipeline
x265_amd.init(0)
lib = x265_amd.api._lib()
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2:   # forced shape of the exact kernel: 8 or 16 waves
    lib.x265amd_debug_sea_shape(0, int(sys.argv[2]))
seq = np.stack([synth.pad_plane(f, 64) for f in synth.make_sequence(1920, 1080, nf + 1, search_range=64, seed=12345)])
pipe = MeTqPipeline(torch.from_numpy(seq).cuda(), ctu=64, search_range=64, nrefs=1, method="sea")
for _ in range(10):
    pipe.motion_search()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    pipe.motion_search()
e1.record()
torch.cuda.synchronize()
lib.x265amd_debug_me_timing(1)
buf = (ctypes.c_float * 4)()
acc = np.zeros(4)
for _ in range(20):
    pipe.motion_search()
    lib.x265amd_debug_me_last_ms(buf)
    acc += np.array(buf[:])
lib.x265amd_debug_me_timing(0)
print(f"search stage {e0.elapsed_time(e1) / 20:.4f} ms back to back; kernels (box, -, bounds, exact + list): {(acc / 20).round(4).tolist()}  mv md5 "
      + __import__('hashlib').md5(x265_amd.mv_to_numpy(pipe.mv).tobytes()).hexdigest()[:8])
