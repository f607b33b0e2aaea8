#!/bin/bash
# round 4, call 8: where the C5 (Gemma-7B) path stands against torch-ROCm bf16 (text-only width probe + the full-depth test with the second
# opinion's own tower); rocprofv3 kernel stats of the 70B decode loop (C4)
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r4c8
timeout 600 python tools/gpu_c5_calibration_probe.py > gpurun_out/r4c8/c5_probe.txt 2>&1
cat gpurun_out/r4c8/c5_probe.txt | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_c2_full_depth_gpu.py -q -k c5 > gpurun_out/r4c8/pytest_c5.txt 2>&1
tail -12 gpurun_out/r4c8/pytest_c5.txt
cd /tmp; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4c8/prof_decode70 -o d70 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 16 meta-llama/Llama-3.3-70B-Instruct > $GRAFT_REPO_ROOT/gpurun_out/r4c8/decode70.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/r4c8/decode70.txt
find gpurun_out/r4c8/prof_decode70 -name "*kernel_stats*" | head -2
f=$(find gpurun_out/r4c8/prof_decode70 -name "*kernel_stats*.csv" | head -1); head -25 "$f" | cut -c1-220
# the raw trace is large: keep only the stats
find gpurun_out/r4c8/prof_decode70 -name "*kernel_trace*" -delete
