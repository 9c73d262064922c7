mkdir -p gpurun_out/r05bh
python -m pytest tests/test_upconv_gpu.py -x -q 2>&1 | tail -2
bash tools/trace_kernel.sh upc_class_sums > gpurun_out/r05bh/class_sums_border_first.txt 2>&1; cat gpurun_out/r05bh/class_sums_border_first.txt
