mkdir -p gpurun_out/r05bh
python -m pytest tests/test_upconv_gpu.py tests/test_mfma_pin_gpu.py -x -q 2>&1 | tail -2
(echo "# upc_class_sums launches of a one-stream step: PCRL_UPC_CLS_WIDE=0 (256 threads per plane) then =1 (1024 where the plane is >= 128 KB)"; PCRL_UPC_CLS_WIDE=0 bash tools/trace_kernel.sh upc_class_sums; echo "---"; PCRL_UPC_CLS_WIDE=1 bash tools/trace_kernel.sh upc_class_sums) > gpurun_out/r05bh/class_sums_ab.txt 2>&1; cat gpurun_out/r05bh/class_sums_ab.txt
bash tools/abn_bench.sh -r 3 -s 20 "PCRL_UPC_CLS_WIDE=0" "PCRL_UPC_CLS_WIDE=1" > gpurun_out/r05bh/step_ab.txt 2>&1; cat gpurun_out/r05bh/step_ab.txt
