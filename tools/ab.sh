#!/bin/bash
# A/B on one box: tools/ab.sh "<python command>" lib1.so lib2.so ...   (three alternating rounds; output to stdout)
cmd=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    X265AMD_LIB_PATH=$PWD/$lib timeout -k 10 120 $cmd 2>&1 | grep -v amdgpu.ids | head -8 || exit 1
  done
done
