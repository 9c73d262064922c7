cd $GRAFT_REPO_ROOT
for V in 1 0 1 0; do echo "== PCRL_IGEMM_VMAJOR=$V"; PCRL_IGEMM_VMAJOR=$V python tools/conv_probe.py --b 192 --what fwd,dgrad --impls 0 --rounds 9 --layers loc256.0,loc256.1,loc512.0,loc512.1,locup256.1 2>&1 | tail -6 | cut -c1-150; done
