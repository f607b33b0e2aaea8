#!/bin/bash
P=$PWD/ultravox_amd/libuvx_prev.so; N=$PWD/ultravox_amd/libuvx.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "norm" 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do
echo prev; UVX_LIB=$P PYTHONPATH=. timeout 200 python tools/gpu_elementwise_probe.py 2>&1 | grep -v amdgpu.ids
echo new; UVX_LIB=$N PYTHONPATH=. timeout 200 python tools/gpu_elementwise_probe.py 2>&1 | grep -v amdgpu.ids
done
