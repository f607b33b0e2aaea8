#!/bin/bash
# Round 2, call 17: attention backward: longest-first block order, cursor iteration, wave-level skip of fully masked tiles.
R=$PWD; OUT=$R/gpurun_out/r2c17; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=15 run tests_attn 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemma_gpu.py tests/test_f32_parity_gpu.py tests/test_lora_gpu.py tests/test_model_gpu.py tests/test_c2_width_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider
TAIL=6 run attn_probe 300 python tools/gpu_attn_bwd_probe.py
bash tools/pmc_attn_bwd.sh 2>&1 | grep -v "^more" | grep "WAVE_CYCLES\|INSTS_VALU\|INSTS_SALU\|GRBM\|WAIT_ANY\|WAIT_INST_ANY\|ACTIVE_INST_ANY "
