cd $GRAFT_REPO_ROOT
for L in "" build/var/libpcrl_idx32.so "" build/var/libpcrl_idx32.so; do echo "== lib=[$L]"; PCRL_LIB=$L python tools/conv2d_probe.py --shapes gather --what fwd 2>&1 | tail -8 | cut -c1-120; done
