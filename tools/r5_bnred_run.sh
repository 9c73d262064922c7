mkdir -p gpurun_out/r05ba
python -m pytest tests/test_dgrad_bnred_gpu.py -x -q > gpurun_out/r05ba/pytest_bnred.log 2>&1; tail -5 gpurun_out/r05ba/pytest_bnred.log
bash tools/abn_bench.sh -r 3 -s 20 "PCRL_DGRAD_BNRED=0" "PCRL_DGRAD_BNRED=1" > gpurun_out/r05ba/step_ab.txt 2>&1; cat gpurun_out/r05ba/step_ab.txt
python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/r05ba/pytest_model.log 2>&1; tail -5 gpurun_out/r05ba/pytest_model.log
