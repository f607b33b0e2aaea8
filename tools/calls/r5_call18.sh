#!/bin/bash
# round 5, call 18: plain projector activations (uvx_config_t.proj_act, ABI 15): the model / kernel / f32-parity / checkpoint test files
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c18; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_f32_parity_gpu.py tests/test_checkpoint_gpu.py tests/test_lora_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest.txt
