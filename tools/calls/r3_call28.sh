#!/bin/bash
P=$PWD/ultravox_amd/libuvx_prev.so; N=$PWD/ultravox_amd/libuvx.so
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_f32_parity_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do
echo prev; UVX_LIB=$P PYTHONPATH=. timeout 200 python tools/gpu_attn_shapes_probe.py 2>&1 | grep -v amdgpu.ids
echo new; UVX_LIB=$N PYTHONPATH=. timeout 200 python tools/gpu_attn_shapes_probe.py 2>&1 | grep -v amdgpu.ids
done
