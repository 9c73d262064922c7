mkdir -p gpurun_out/r05bc
python -m pytest tests/test_upconv_gpu.py -x -q > gpurun_out/r05bc/pytest_upconv.log 2>&1; tail -3 gpurun_out/r05bc/pytest_upconv.log
bash tools/abn_bench.sh -r 3 -s 20 "PCRL_UPC_PACK_TILED=0" "PCRL_UPC_PACK_TILED=1" > gpurun_out/r05bc/pack_ab.txt 2>&1; cat gpurun_out/r05bc/pack_ab.txt
