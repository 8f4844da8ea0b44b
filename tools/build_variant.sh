#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." [file.hip]: the library with ONE translation unit (default conv.hip) rebuilt with
# extra flags -> tools/bin/libacktr_NAME.so, for same-box A/B runs (tools/ab_update.py tools/bin/libacktr_A.so ...)
set -e
NAME=$1; EXTRA=$2; SRC=${3:-conv.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/pytorch-a2c-ppo-acktr-gail_amd/csrc
mkdir -p $ROOT/tools/bin
OBJ=$ROOT/tools/bin/${SRC%.hip}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -I $ROOT/include -I $CS -I /opt/rocm/include $EXTRA -c $CS/$SRC -o $OBJ
OTHERS=$(ls $CS/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libacktr_$NAME.so $OBJ $OTHERS -L/opt/rocm/lib -lrocsolver -lrocblas -lpthread -Wl,-rpath,/opt/rocm/lib
echo built tools/bin/libacktr_$NAME.so
