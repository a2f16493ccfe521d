#!/bin/bash
# Everything profiles/ holds, in one gpurun call: bench lines of every workload, rocprofv3 kernel stats of the
# default bench, PMC traffic.  Each GPU step under its own time limit; stops at the first failure.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python bench.py > $OUT/bench_1080p.json 2> $OUT/bench_1080p.err
echo "1080p done"
timeout -k 10 300 python bench.py --workload 4k --no-cpu-baseline > $OUT/bench_4k.json 2> $OUT/bench_4k.err
echo "4k done"
timeout -k 10 300 python bench.py --workload 8k4ref --no-cpu-baseline > $OUT/bench_8k4ref.json 2> $OUT/bench_8k4ref.err
echo "8k done"
timeout -k 10 300 python bench.py --workload tq1080p --tu 32 > $OUT/bench_tq1080p.json 2> $OUT/bench_tq.err
timeout -k 10 300 python bench.py --workload gop1080p > $OUT/bench_gop1080p.json 2> $OUT/bench_gop.err
echo "tq, gop done"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace done"
cd $ROOT
bash tools/run_pmc.sh 1080p 3 1 > $OUT/pmc.log 2>&1
cp $ROOT/gpurun_out/pmc_1080p/pmc_traffic.json $OUT/pmc_traffic.json
echo "pmc done"
