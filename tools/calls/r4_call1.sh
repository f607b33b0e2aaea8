#!/bin/bash
# round 4, call 1: four-wave hand-scheduled GEMM (variants 43..45): bit-identity vs the eight-wave kernel, race screen, cold probe
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r4c1
timeout 600 python tools/gpu_gemm_a4_check.py 43 > gpurun_out/r4c1/check43.txt 2>&1
tail -15 gpurun_out/r4c1/check43.txt
timeout 300 python tools/gpu_gemm_a4_check.py 44,45 > gpurun_out/r4c1/check44_45.txt 2>&1
tail -3 gpurun_out/r4c1/check44_45.txt
timeout 600 python tools/gpu_gemm_cold_probe.py 31,33,43,44,45 > gpurun_out/r4c1/cold_llm.txt 2>&1
cat gpurun_out/r4c1/cold_llm.txt
