#!/bin/bash
# round 4, call 7: new parity asserts (full-depth bars + GPU torch-bf16 calibration, C5 full depth), KL + LLM-LoRA fixture, cached forward with
# [B, Tn, V] logits; split-K tail feasibility probe; baseline bench line of this box
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r4c7
timeout 300 python -m pytest tests/test_lora_gpu.py tests/test_generate_gpu.py tests/test_kl_gpu.py -x -q > gpurun_out/r4c7/pytest_api.txt 2>&1
tail -5 gpurun_out/r4c7/pytest_api.txt
timeout 900 python -m pytest tests/test_c2_full_depth_gpu.py -q > gpurun_out/r4c7/pytest_full_depth.txt 2>&1
tail -30 gpurun_out/r4c7/pytest_full_depth.txt
timeout 400 python tools/gpu_gemm_tailsplit_probe.py > gpurun_out/r4c7/tailsplit.txt 2>&1
cat gpurun_out/r4c7/tailsplit.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4c7/bench.json 2> gpurun_out/r4c7/bench.err
cat gpurun_out/r4c7/bench.json
