#!/bin/bash
# PCIe-inclusive rate of the C++ host loop: 1080p (padded to 1088), 16 segments, 5 frames
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from x265_amd import synth
W, H, T = 1920, 1088, 9
y, u, v = synth.make_sequence(W, H, T, search_range=64, seed=12345, chroma=True)
np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1).tofile('/tmp/host_bench.yuv')
PY
for S in 1 4 16; do timeout -k 10 120 x265_amd/lib/x265amd_encode_p --bench /tmp/host_bench.yuv 1920 1088 9 32 64 $S; done
