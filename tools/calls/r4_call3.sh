#!/bin/bash
# round 4, call 3: eight-wave hand-scheduled loop (variants 49..51) - bit-identity, race screen, per-K-tile cost with its probe builds, cold probe
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r4c3
timeout 900 python tools/gpu_gemm_a4_check.py 49,50,51,43 > gpurun_out/r4c3/check.txt 2>&1
grep -c OK gpurun_out/r4c3/check.txt; grep -v " OK" gpurun_out/r4c3/check.txt | tail -8
timeout 600 python tools/gpu_gemm_ktile_probe.py 31,43,49,50,51,52,53,54 > gpurun_out/r4c3/ktile.txt 2>&1
cat gpurun_out/r4c3/ktile.txt
timeout 600 python tools/gpu_gemm_cold_probe.py 31,33,43,49,50,51 > gpurun_out/r4c3/cold_llm.txt 2>&1
cat gpurun_out/r4c3/cold_llm.txt
