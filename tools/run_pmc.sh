#!/bin/bash
# PMC traffic passes for bench.py (run on the GPU box through gpurun): separate FETCH_SIZE and WRITE_SIZE runs,
# no tracing domains beside them.  usage: tools/run_pmc.sh <workload> <steps> <warmup> [key] [extra bench args]
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-1080p}; STEPS=${2:-3}; WARM=${3:-1}; KEY=${4:-$WL}; shift 4 2>/dev/null || true
OUT=$ROOT/gpurun_out/pmc_$KEY
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline "$@" > $OUT/write.log 2>&1
python3 $ROOT/tools/pmc_traffic.py $KEY $((STEPS + WARM)) $OUT/fetch $OUT/write ${PMC_JSON:-$OUT/pmc_traffic.json}
