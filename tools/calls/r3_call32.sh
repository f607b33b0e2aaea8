#!/bin/bash
P=$PWD/ultravox_amd/libuvx_prev.so; N=$PWD/ultravox_amd/libuvx.so; F=$PWD/ultravox_amd/libuvx_pf.so
for i in 1 2; do
for v in prev:$P mv2:$N t512_mv1:$F; do
echo ${v%%:*}; UVX_LIB=${v#*:} PYTHONPATH=. timeout 200 python tools/gpu_elementwise_probe.py 2>&1 | grep rmsnorm_bwd
done; done
