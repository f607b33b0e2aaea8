#!/bin/bash
# round 5, call 5: weight row-stride probe (power-of-two ldb vs padded); C4 full-depth parity test again (the flash restatement now defines
# a fully masked query row as 0 instead of 0 / 0)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c5; mkdir -p $O
timeout 500 python tools/gpu_gemm_ldb_probe.py > $O/ldb_probe.txt 2>&1; grep -v amdgpu.ids $O/ldb_probe.txt
timeout 900 python -m pytest tests/test_c4_full_depth_gpu.py -q -x -p no:cacheprovider > $O/pytest_c4_full_depth.txt 2>&1; tail -5 $O/pytest_c4_full_depth.txt
cat gpurun_out/parity/c4_full_depth.json
