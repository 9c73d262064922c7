cd $GRAFT_REPO_ROOT
for B in 128 64; do echo "== PCRL_GATHER_BN=$B conv2d_probe"; PCRL_GATHER_BN=$B python tools/conv2d_probe.py --shapes gather --what fwd 2>&1 | tail -8 | cut -c1-120; done
for B in 128 64; do echo "== PCRL_GATHER_BN=$B local 3D"; PCRL_GATHER_BN=$B python tools/conv_probe.py --b 192 --what fwd --impls 0 --rounds 7 --layers loc256.0,loc256.1,loc512.0,loc512.1,locup256.1 2>&1 | tail -6 | cut -c1-110; done
for i in 1 2; do for B in 128 64; do echo "== PCRL_GATHER_BN=$B 2D step"; PCRL_GATHER_BN=$B python tools/bench_2d.py --steps 6 --warmup 3 --no-roofline 2>&1 | tail -1 | cut -c1-110; done; done
