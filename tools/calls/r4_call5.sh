#!/bin/bash
# round 4, call 5: ping-pong eight-wave loop (variant 55; 56..58 timing probes): bit-identity, race screen, per-K-tile cost, cold probe
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r4c5
timeout 600 python tools/gpu_gemm_a4_check.py 55 > gpurun_out/r4c5/check.txt 2>&1
grep -c OK gpurun_out/r4c5/check.txt; grep -v " OK" gpurun_out/r4c5/check.txt | tail -8
timeout 600 python tools/gpu_gemm_ktile_probe.py 31,49,55,56,57,58 > gpurun_out/r4c5/ktile.txt 2>&1
cat gpurun_out/r4c5/ktile.txt
timeout 600 python tools/gpu_gemm_cold_probe.py 31,33,49,55 > gpurun_out/r4c5/cold_llm.txt 2>&1
cat gpurun_out/r4c5/cold_llm.txt
