#!/bin/bash
# Build a variant of the current library with one source recompiled under extra flags (same-box A/B of compile-time knobs):
#   tools/build_variant.sh <name> <file.hip> <flags...>   ->  build/var/libpcrl_<name>.so ; use with PCRL_LIB=build/var/libpcrl_<name>.so
NAME=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc "$@" -c $R/pcrlv2_amd/csrc/$SRC -o $R/build/var/${NAME}_$(basename $SRC .hip).o || exit 1
OBJS=$(ls $R/build/obj/*.o | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build/var/libpcrl_$NAME.so $OBJS $R/build/var/${NAME}_$(basename $SRC .hip).o && echo built build/var/libpcrl_$NAME.so
