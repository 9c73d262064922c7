#!/bin/bash
# Build the library of a previous revision (default HEAD) next to the current one for same-box A/B runs:
#   tools/build_old_lib.sh [rev]  ->  build/old/libpcrl_hip_old.so ; use with PCRL_LIB=build/old/libpcrl_hip_old.so
REV=${1:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf $R/build/old && mkdir -p $R/build/old/src/pcrlv2_amd/csrc $R/build/old/src/include
git -C $R archive $REV pcrlv2_amd/csrc include | tar -x -C $R/build/old/src
cd $R/build/old
for f in src/pcrlv2_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -c $f -o $(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libpcrl_hip_old.so *.o && echo built $R/build/old/libpcrl_hip_old.so
