#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
PYTHONPATH=. timeout 300 python tools/gpu_attn_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_attn_timeline_c29.txt
PYTHONPATH=. timeout 200 python tools/gpu_attn_shapes_probe.py 2>&1 | grep -v amdgpu.ids
