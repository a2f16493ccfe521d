#!/bin/bash
# Two libraries for tools/ab.sh: x265_amd/lib/var_old.so = HEAD's <file>.hip with today's other objects, var_new.so = the
# working tree's build.   tools/ab_build.sh me_sea      (run `make -C x265_amd/csrc` first)
set -e
F=${1:-me_sea}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/x265_amd/csrc $T/include
cp $ROOT/include/*.h $T/include/
git -C $ROOT archive HEAD x265_amd/csrc | tar -x -C $T
(cd $T/x265_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -c $F.hip -o $T/$F.o)
cd $ROOT/x265_amd/lib
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o var_old.so $(ls *.o | grep -v "^$F.o") $T/$F.o
cp libx265amd.so var_new.so
rm -rf $T
ls -la var_old.so var_new.so
