#!/bin/bash
# Same-box probe of the weight-gradient kernels under the compile-time ablations of wgrad_brick.hip (tools/build_variant.sh abl<N> wgrad_brick.hip -DWB_ABL=<N>):
#   tools/wgrad_abl.sh <outfile> <variant> ...      (variant "base" = the in-tree library)
OUT=$1; shift
LAYERS=${LAYERS:-down64.1,up64.1,down128.1,up128.0,up128.1,up256.0,up256.1}
mkdir -p $(dirname $OUT); : > $OUT
for v in "$@"; do
  echo "== $v" >> $OUT
  if [ "$v" = base ]; then unset PCRL_LIB; else export PCRL_LIB=build/var/libpcrl_$v.so; fi
  python tools/conv_probe.py --what wgrad --wimpls 0 --rounds 7 --layers $LAYERS >> $OUT 2>&1
done
cat $OUT
