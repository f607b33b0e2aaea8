#!/bin/bash
P=$PWD/ultravox_amd/libuvx_prev.so; N=$PWD/ultravox_amd/libuvx.so; F=$PWD/ultravox_amd/libuvx_pf.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "norm" 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do
for v in prev:$P mv2:$N mv2_prefetch:$F; do
echo ${v%%:*}; UVX_LIB=${v#*:} PYTHONPATH=. timeout 200 python tools/gpu_elementwise_probe.py 2>&1 | grep rmsnorm_bwd
done; done
