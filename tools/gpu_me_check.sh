#!/bin/bash
# GPU box: the search's parity tests, then a short bench line and (optionally) the per-CTU statistics.
# usage: tools/gpu_me_check.sh TAG [stats]
tag=$1
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "me_ or sea or no_alloc or gop or pipeline or fullsize" > gpurun_out/${tag}_t.log 2>&1
rc=$?
echo rc=$rc >> gpurun_out/${tag}_t.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 100 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit $?
if [ "$2" = stats ]; then
  timeout -k 10 200 python tools/sea_stats.py 16 > gpurun_out/${tag}_stats.txt 2>&1 || exit $?
fi
