cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for L in "" "build/var/libpcrl_narrow_nt.so"; do
    echo "== lib=[$L]"; PCRL_LIB=$L python tools/bench_2d.py --steps 6 --warmup 3 --no-roofline 2>&1 | tail -1 | cut -c1-120
  done
done
