#!/bin/bash
# round 3, call 1: the new parity tests (full depth C2 / C3, f32 mode at C2 width, reference-run encoder fixtures) + a baseline bench line
mkdir -p gpurun_out
python -m pytest tests/test_c2_full_depth_gpu.py tests/test_f32_parity_gpu.py -m gpu -q -k "full_depth or c2_width_f32 or reference_encoder_forward or its_own_tower" --durations=8 > gpurun_out/r3c1_tests.log 2>&1
tail -25 gpurun_out/r3c1_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --gemm-table gpurun_out/r3c1_gemm_table.txt > gpurun_out/r3c1_bench.json 2> gpurun_out/r3c1_bench.err
cat gpurun_out/r3c1_bench.json | cut -c1-600
