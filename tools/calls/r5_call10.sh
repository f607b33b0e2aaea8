#!/bin/bash
# round 5, call 10: the three test files again (cache rows compared up to cur_len; keep_params read back by name), the wav2vec2 tower stage probe
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c10; mkdir -p $O
timeout 600 python -m pytest tests/test_lora_gpu.py tests/test_checkpoint_gpu.py tests/test_generate_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest.txt
timeout 400 python tools/gpu_c5_tower_stage_probe.py 1,4,12,24 10 > $O/c5_tower_stage_probe.txt 2>&1; grep -v amdgpu.ids $O/c5_tower_stage_probe.txt | tail -8
