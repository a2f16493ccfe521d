#!/bin/bash
# GPU box: the transform path's parity tests, then the tq1080p bench line at every TU size.
tag=$1
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "tq or intra or gop or dropin or idct or pipeline or cpp_host" > gpurun_out/${tag}_t.log 2>&1
rc=$?
echo rc=$rc >> gpurun_out/${tag}_t.log
[ $rc -eq 0 ] || exit $rc
for tu in 32 16 8 4; do
  timeout -k 10 100 python bench.py --workload tq1080p --tu $tu --lean > gpurun_out/${tag}_tq${tu}.json 2> gpurun_out/${tag}_tq${tu}.err || exit $?
done
