#!/bin/bash
# A/B builds of ONE translation unit of libtgn_pointops.so: tools/ab_build.sh <name> <file.hip> [-Dflags...]  ->  tools/_bin/ab/libtgn_<name>.so
# (run the product with TGN_LIB_PATH=tools/_bin/ab/libtgn_<name>.so)
set -e
name=$1; src=$2; shift 2
C=toothgroupnetwork_amd/csrc
mkdir -p tools/_bin/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -munsafe-fp-atomics -ffp-contract=off "$@" -c $C/$src -o tools/_bin/ab/${name}_${src%.hip}.o
objs=$(ls $C/_obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/_bin/ab/${name}_${src%.hip}.o -o tools/_bin/ab/libtgn_${name}.so
echo built tools/_bin/ab/libtgn_${name}.so
