cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops2d_gpu.py tests/test_model2d_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -2
for V in 1 0 1 0; do echo "== PCRL_WGRAD2D_ROW3=$V"; PCRL_WGRAD2D_ROW3=$V python tools/conv2d_probe.py --shapes gather --what wgrad 2>&1 | tail -8 | cut -c1-120; done
for i in 1 2; do for V in 1 0; do echo "== PCRL_WGRAD2D_ROW3=$V 2D step"; PCRL_WGRAD2D_ROW3=$V python tools/bench_2d.py --steps 6 --warmup 3 --no-roofline 2>&1 | tail -1 | cut -c1-110; done; done
