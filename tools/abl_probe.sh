cd $GRAFT_REPO_ROOT
for V in 1 0 1 0; do echo "== PCRL_C1_PERSIST=$V"; PCRL_C1_PERSIST=$V python tools/c1_probe.py 2>&1 | tail -3; done
python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "c1" 2>&1 | grep -E "passed|failed" | tail -2
