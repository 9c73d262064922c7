mkdir -p gpurun_out/r05bk
python -m pytest tests/test_dgrad_bnred_gpu.py -x -q 2>&1 | tail -2
bash tools/step_launch_table.sh r05bk > gpurun_out/r05bk_slt.log 2>&1
grep "brick16_conv_kernel<32" gpurun_out/r05bk_launch_groups.txt | cut -c1-130
ABL_SETS=no_bnred python tools/double_ablation.py --steps 12 --rounds 5 > gpurun_out/r05bk/bnred_step_ab.txt 2>&1; tail -2 gpurun_out/r05bk/bnred_step_ab.txt
