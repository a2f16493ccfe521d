#!/bin/bash
# per-kernel times and the copy / kernel timeline of the C++ host loop (encode_p --bench) under rocprofv3:
#   tools/host_prof.sh <nseg> [mode,streams [pipe]]     then: python tools/host_timeline.py gpurun_out/host_prof 80
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
S=${1:-16}
[ -n "$2" ] && export X265AMD_BENCH_ONLY=$2
[ -n "$3" ] && export X265AMD_BENCH_PIPE=1
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from x265_amd import synth
W, H, T = 1920, 1088, 9
y, u, v = synth.make_sequence(W, H, T, search_range=64, seed=12345, chroma=True)
np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1).tofile('/tmp/host_bench.yuv')
PY
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/host_prof
rm -rf $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT -o hp --output-format csv -- x265_amd/lib/x265amd_encode_p --bench /tmp/host_bench.yuv 1920 1088 9 32 64 $S > $ROOT/gpurun_out/host_prof.log 2>&1
grep '^{' $ROOT/gpurun_out/host_prof.log | cut -c 40-75,150-260
