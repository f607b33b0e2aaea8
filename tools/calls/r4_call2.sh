#!/bin/bash
# round 4, call 2: per-K-tile cost of the four-wave loop and of its probe builds (no DMA / no DMA + reads / no MFMAs) next to the eight-wave kernel
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r4c2
timeout 600 python tools/gpu_gemm_ktile_probe.py 31,43,44,45,46,47,48 > gpurun_out/r4c2/ktile.txt 2>&1
cat gpurun_out/r4c2/ktile.txt
