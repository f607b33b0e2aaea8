#!/bin/bash
# round 6, call 23: the wav2vec2 tower under apply_lora (uvx_wav2vec2_fwd_train / uvx_wav2vec2_bwd, ABI 17): its tests, the rest of the wav2vec2 file (the forward
# was refactored into attention / feed-forward branch helpers), the LoRA file (helpers moved to lora.hip)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c23; mkdir -p $O
timeout 900 python -m pytest tests/test_wav2vec2_gpu.py -q -x -k "lora" 2>&1 | tail -25 | tee $O/pytest_new.txt
timeout 1500 python -m pytest tests/test_wav2vec2_gpu.py tests/test_lora_gpu.py tests/test_gemma_gpu.py -q 2>&1 | tail -8 | tee $O/pytest.txt
