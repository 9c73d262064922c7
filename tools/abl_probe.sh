cd $GRAFT_REPO_ROOT
for V in 0 1 0 1; do echo "== PCRL_NARROW_FIRST=$V"; PCRL_NARROW_FIRST=$V python tools/conv2d_probe.py --shapes brick --what fwd --only 2 2>&1 | tail -1 | cut -c1-120;  PCRL_NARROW_FIRST=$V python tools/bench_2d.py --steps 6 --warmup 3 --no-roofline 2>&1 | tail -1 | cut -c1-110; done
