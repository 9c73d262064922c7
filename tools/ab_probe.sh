#!/bin/bash
# tools/ab_probe.sh "<lib1> <lib2> ..." <rounds> <conv_probe args...>: run tools/conv_probe.py once per library and round, alternating
LIBS="$1"; R=$2; shift 2
for i in $(seq 1 $R); do for L in $LIBS; do
  if [ "$L" = cur ]; then unset PCRL_LIB; else export PCRL_LIB=$L; fi
  echo "== $L"; timeout 600 python tools/conv_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-150
done; done
