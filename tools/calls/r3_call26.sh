#!/bin/bash
P=$PWD/ultravox_amd/libuvx_prev.so; N=$PWD/ultravox_amd/libuvx.so
for i in 1 2; do
echo prev; UVX_LIB=$P PYTHONPATH=. timeout 200 python tools/gpu_attn_shapes_probe.py 2>&1 | grep -v amdgpu.ids
echo new; UVX_LIB=$N PYTHONPATH=. timeout 200 python tools/gpu_attn_shapes_probe.py 2>&1 | grep -v amdgpu.ids
done
