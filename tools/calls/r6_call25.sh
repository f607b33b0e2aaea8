#!/bin/bash
# round 6, call 25: ABI 18 (LoRA adapters on the MLP's linears) - the LoRA GPU tests, then the ABI / model tests that touch the changed structs
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6c25; mkdir -p $O
timeout 900 python -m pytest tests/test_lora_gpu.py -x -q -m gpu > $O/lora_tests.log 2>&1; echo "lora rc $?"; tail -15 $O/lora_tests.log
timeout 600 python -m pytest tests/test_wav2vec2_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/model_tests.log 2>&1; echo "model rc $?"; tail -5 $O/model_tests.log
