#!/bin/bash
# round 4, call 6: where the K-tile's cycles go - PMC (clock, MFMA busy, wave split) on the 160-tile shape + per-K-tile cost with 160 CUs busy
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r4c6
timeout 300 python tools/gpu_gemm_ktile_probe.py 31,33,49,55,56,57,58 2560 > gpurun_out/r4c6/ktile_160tiles.txt 2>&1
cat gpurun_out/r4c6/ktile_160tiles.txt
VARS="31 49 55 56 57 58" SHAPE="2560 4096 8192" OUT=gpurun_out/r4c6/pmc160 timeout 1500 bash tools/pmc_gemm_variants.sh
VARS="31 55" SHAPE="4096 4096 8192" OUT=gpurun_out/r4c6/pmc256 timeout 600 bash tools/pmc_gemm_variants.sh
