#!/bin/bash
# a variant libd3hip.so with extra -D flags for ONE source (the other objects are the default build's): tools/probes/build_variant.sh <name> <source.hip> <flags...>
# -> d3net_amd/lib/variants/libd3hip_<name>.so (git-ignored with lib/; travels to the GPU box); load with D3_SO=<path> in the probe tools
set -e
NAME=$1; SRC=$2; shift; shift
ROOT=$(cd $(dirname $0)/../.. && pwd)
python -m d3net_amd.build > /dev/null
mkdir -p $ROOT/d3net_amd/lib/variants
OBJ=$ROOT/d3net_amd/lib/variants/${NAME}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I$ROOT/include "$@" -c $ROOT/d3net_amd/csrc/$SRC -o $OBJ
OBJS=$(ls $ROOT/d3net_amd/build/*.o | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o $ROOT/d3net_amd/lib/variants/libd3hip_$NAME.so $OBJS $OBJ
echo built $ROOT/d3net_amd/lib/variants/libd3hip_$NAME.so
