mkdir -p gpurun_out/r05bb
python tools/double_ablation.py > gpurun_out/r05bb/double_ablation.txt 2>&1; tail -8 gpurun_out/r05bb/double_ablation.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r05bb/pytest_gpu.log 2>&1; tail -5 gpurun_out/r05bb/pytest_gpu.log
