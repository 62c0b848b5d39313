#!/bin/bash
# A/B builds of libpcc_geo_hip.so: tools/build_variant.sh <name> <source.hip> "<extra -D flags>"
# -> build_ab/lib<name>.so (select with PCC_GEO_LIB=build_ab/lib<name>.so); all other objects come from the in-tree build.
set -e
NAME=$1; SRC=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/pcc_geo_cnn_v2_amd/csrc
mkdir -p $R/build_ab
make -s -j8 -C $C
BASE=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $FLAGS -c $C/$SRC -o $R/build_ab/${BASE}_$NAME.o
OBJS=$(ls $C/*.o | grep -v "/${BASE}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ab/lib$NAME.so $OBJS $R/build_ab/${BASE}_$NAME.o -lpthread
echo built build_ab/lib$NAME.so
