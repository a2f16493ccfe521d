#!/bin/bash
# End-to-end rate of the C++ intra encoder (x265amd_encode_i: upload, GPU pixel path, download, syntax on host
# threads; two slots take turns, so the GPU side of a batch overlaps the syntax of the batch before) on the GPU box;
# <batch> <threads> <dumps>: dumps = 0 writes the bitstream only:  bash tools/encode_i_bench.sh > gpurun_out/encode_i_bench.txt
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
python3 - "$T" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from x265_amd import synth
synth.make_intra_probe(1920, 1088, 8).tofile(sys.argv[1] + "/p1080.yuv")
PY
python3 - "$T" <<'PY'
import sys, numpy as np
a = np.fromfile(sys.argv[1] + "/p1080.yuv", np.uint8)
np.tile(a, 4).tofile(sys.argv[1] + "/in.yuv")   # 32 frames
np.tile(a, 16).tofile(sys.argv[1] + "/in128.yuv")   # 128 frames
PY
for cfg in "1 1 1" "8 8 1" "16 16 1" "1 1 0" "4 4 0" "8 8 0" "16 16 0"; do
    set -- $cfg
    timeout -k 10 120 x265_amd/lib/x265amd_encode_i "$T/in.yuv" 1920 1088 32 32 32 "$T/out" "$1" "$2" "$3" | tail -1
done
for cfg in "8 16 0" "16 16 0"; do
    set -- $cfg
    timeout -k 10 120 x265_amd/lib/x265amd_encode_i "$T/in128.yuv" 1920 1088 128 32 32 "$T/out" "$1" "$2" "$3" | tail -1
done
rm -rf "$T"
