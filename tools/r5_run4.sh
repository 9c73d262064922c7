mkdir -p gpurun_out/r05be
python -m pytest tests/test_dgrad_bnred_gpu.py tests/test_model2d_gpu.py -x -q > gpurun_out/r05be/pytest_2d.log 2>&1; tail -4 gpurun_out/r05be/pytest_2d.log
for i in 1 2 3; do
  PCRL_DGRAD_BNRED=0 python tools/bench_2d.py --steps 8 --warmup 4 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r05be/c5_ab_off.txt
  PCRL_DGRAD_BNRED=1 python tools/bench_2d.py --steps 8 --warmup 4 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r05be/c5_ab_on.txt
done
echo OFF; cat gpurun_out/r05be/c5_ab_off.txt; echo ON; cat gpurun_out/r05be/c5_ab_on.txt
