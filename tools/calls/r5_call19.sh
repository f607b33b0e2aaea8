#!/bin/bash
# round 5, call 19: the wav2vec2 layer-norm (-lv60) family on the GPU + the files its changes touch
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c19; mkdir -p $O
timeout 900 python -m pytest tests/test_wav2vec2_gpu.py tests/test_gemma_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest.txt
