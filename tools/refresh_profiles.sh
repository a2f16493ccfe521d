#!/bin/bash
# Everything profiles/ holds for a round, from the GPU box (through gpurun), in parts so that each call stays well
# inside gpurun's limit:  tools/refresh_profiles.sh bench | trace | pmc | sq
# Results under gpurun_out/prof/; tools/collect_profiles.py copies the summaries into profiles/ afterwards.
# Each GPU step runs under its own time limit; the script stops at the first failure.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd $ROOT
case "$1" in
bench)
  timeout -k 10 500 python bench.py > $OUT/bench_1080p.json 2> $OUT/bench_1080p.err; echo "1080p done"
  # the other search configurations in full too: statistics pass, CPU sample, in-run parity
  timeout -k 10 400 python bench.py --workload 4k > $OUT/bench_4k.json 2> $OUT/bench_4k.err; echo "4k done"
  timeout -k 10 500 python bench.py --workload 8k4ref --steps 20 --warmup 5 > $OUT/bench_8k4ref.json 2> $OUT/bench_8k4ref.err; echo "8k4ref done"
  timeout -k 10 300 python bench.py --workload gop1080p --no-cpu-baseline > $OUT/bench_gop1080p.json 2> $OUT/bench_gop1080p.err; echo "gop1080p done"
  for TU in 32 16 8 4; do
    timeout -k 10 200 python bench.py --workload tq1080p --tu $TU --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_tq1080p_tu$TU.json 2> $OUT/bench_tq_tu$TU.err
  done
  echo "tq done"
  timeout -k 10 300 python bench.py --workload intra1080p > $OUT/bench_intra1080p.json 2> $OUT/bench_intra1080p.err; echo "intra1080p done"
  timeout -k 10 200 python tools/pu_bench.py > $OUT/pu_bench.txt 2>&1; echo "pu done"
  timeout -k 10 300 python tools/intra_bench.py > $OUT/intra_bench.txt 2>&1; echo "intra done"
  timeout -k 10 100 python tools/tq_trace.py 32 16 > $OUT/tq_trace.txt 2>&1; echo "tq trace done"
  timeout -k 10 400 bash tools/encode_i_bench.sh > $OUT/encode_i_bench.txt 2>&1; echo "encode_i done"
  timeout -k 10 100 python tools/sea_stats.py 16 > $OUT/sea_stats.txt 2>&1; echo "sea stats done"
  timeout -k 10 100 python tools/noise_split.py > $OUT/noise_split.txt 2>&1; echo "noise split done"
  (timeout -k 10 100 python tools/intra_stats.py 1 && timeout -k 10 100 python tools/intra_stats.py 16) > $OUT/intra_stats.txt 2>&1; echo "intra stats done"
  timeout -k 10 300 bash tools/host_bench.sh > $OUT/host_pcie.txt 2>&1; echo "host bench done"
  timeout -k 10 100 python tools/gop_steps.py 8 pingpong > $OUT/gop_steps.txt 2>&1; echo "gop steps done"
  ;;
trace)
  cd /tmp && export TMPDIR=/tmp
  for WL in 1080p 4k 8k4ref gop1080p tq1080p intra1080p; do
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$WL --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace_$WL.log 2>&1
    echo "trace $WL done"
  done
  ;;
pmc)
  export PMC_JSON=$OUT/pmc_traffic.json
  for WL in 1080p 4k 8k4ref gop1080p tq1080p; do
    bash tools/run_pmc.sh $WL 3 1 $WL > $OUT/pmc_$WL.log 2>&1; echo "pmc $WL done"
  done
  bash tools/run_pmc.sh intra1080p 2 1 intra1080p > $OUT/pmc_intra1080p.log 2>&1; echo "pmc intra1080p done"
  for TU in 16 8 4; do
    bash tools/run_pmc.sh tq1080p 3 1 tq1080p_tu$TU --tu $TU > $OUT/pmc_tq_tu$TU.log 2>&1; echo "pmc tq tu$TU done"
  done
  ;;
sq)
  bash tools/run_pmc_sq.sh 1080p 2 1 > $OUT/sq_1080p.log 2>&1; echo "sq 1080p done"
  bash tools/run_pmc_tq.sh tq1080p 3 1 > $OUT/sq_tq.log 2>&1; echo "sq tq done"
  bash tools/run_pmc_sq.sh intra1080p 2 1 > $OUT/sq_intra1080p.log 2>&1; echo "sq intra1080p done"
  ;;
*) echo "usage: $0 bench|trace|pmc|sq"; exit 2;;
esac
