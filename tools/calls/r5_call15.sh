#!/bin/bash
# round 5, call 15: generate tests on the build that keeps the fused reduce + norm off at the one-wave-per-row widths (512 / 1024 / 2048)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c15; mkdir -p $O
timeout 900 python -m pytest tests/test_generate_gpu.py tests/test_gemma_gpu.py tests/test_qwen_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest.txt
