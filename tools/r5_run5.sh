mkdir -p gpurun_out/r05bf
python -m pytest tests/test_dgrad_bnred_gpu.py -x -q 2>&1 | tail -2
python tools/bnred_probe.py > gpurun_out/r05bf/bnred_probe.txt 2>&1; cat gpurun_out/r05bf/bnred_probe.txt
ABL_SETS=no_bnred python tools/double_ablation.py --steps 12 --rounds 5 > gpurun_out/r05bf/bnred_step_ab.txt 2>&1; tail -2 gpurun_out/r05bf/bnred_step_ab.txt
