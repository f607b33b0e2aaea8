#!/bin/bash
# round 6, call 22: the Mistral family (a Llama block with the sliding window on every layer; uvx_llm_weights_t.layer_local now flavour-independent):
# its tests, then everything that touches the windowed kernels / generate / the configs
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c22; mkdir -p $O
timeout 900 python -m pytest tests/test_mistral_gpu.py -q -x 2>&1 | tail -15 | tee $O/pytest_new.txt
timeout 1500 python -m pytest tests/test_gemma3_gpu.py tests/test_generate_gpu.py tests/test_qwen_gpu.py tests/test_gemma_gpu.py tests/test_model_gpu.py -q 2>&1 | tail -8 | tee $O/pytest.txt
