#!/bin/bash
# A/B builds of libpcc_geo_hip.so: tools/build_variant.sh <name> <source> "<extra -D flags>" [object-to-replace]
#   <source>: a file name under csrc/ or a path (e.g. an old revision exported with `git show`);
#   conv_mfma.hip is built in parts: pass "-DPCC_PART=k" and conv_mfma_pk.o as the object to replace.
# -> build_ab/lib<name>.so (select with PCC_GEO_LIB=build_ab/lib<name>.so); all other objects come from the in-tree build.
set -e
NAME=$1; SRC=$2; FLAGS=$3; REPL=$4
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/pcc_geo_cnn_v2_amd/csrc
mkdir -p $R/build_ab
make -s -j8 -C $C
BASE=$(basename $SRC .hip)
[ -f "$SRC" ] && SRCP=$SRC || SRCP=$C/$SRC
REPL=${REPL:-$BASE.o}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -I$C $FLAGS -c $SRCP -o $R/build_ab/${BASE}_$NAME.o
OBJS=$(ls $C/*.o | grep -v "/${REPL}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ab/lib$NAME.so $OBJS $R/build_ab/${BASE}_$NAME.o -lpthread
echo built build_ab/lib$NAME.so
