#!/bin/bash
# round 5, call 32: the C2 training step's GEMM shapes (LLM and encoder / projector) on cold weights: production tiles, the picker's choice and hipBLASLt
# (torch.matmul) - tools/gpu_gemm_cold_probe.py on the final build
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c32; mkdir -p $O
timeout 600 python tools/gpu_gemm_cold_probe.py 31,32,33,34 > $O/gemm_cold_llm.txt 2>&1; grep -v amdgpu.ids $O/gemm_cold_llm.txt
timeout 600 python tools/gpu_gemm_cold_probe.py 31,32,33,34 enc > $O/gemm_cold_enc.txt 2>&1; grep -v amdgpu.ids $O/gemm_cold_enc.txt
