#!/bin/bash
# round 5, call 4: stream-K (probes build, variant 41 = 160 x 256) on the prefill's gate|up GEMM - 448 tiles = 1.75 rounds of the 256 CUs -
# against the data-parallel tile; the C4 full-depth parity test (80-layer Llama-3.3-70B generate vs torch bf16, teacher-forced)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c4; mkdir -p $O
UVX_LIB=$GRAFT_REPO_ROOT/ultravox_amd/libuvx_probes.so timeout 400 python tools/gpu_gemm_splitk_probe.py all 316,632 33,41,42 > $O/streamk_probe.txt 2>&1; grep -v amdgpu.ids $O/streamk_probe.txt | grep -v "s2=" | head -60
timeout 900 python -m pytest tests/test_c4_full_depth_gpu.py -q -x -p no:cacheprovider > $O/pytest_c4_full_depth.txt 2>&1; tail -15 $O/pytest_c4_full_depth.txt
cat gpurun_out/parity/c4_full_depth.json
