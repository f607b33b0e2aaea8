#!/bin/bash
# Round 2, call 20: the HIP path against the reference-forward / reference-generate fixtures.
R=$PWD; OUT=$R/gpurun_out/r2c20; mkdir -p $OUT; export PYTHONPATH=$R
timeout 300 python -m pytest tests/test_f32_parity_gpu.py tests/test_generate_gpu.py tests/test_kl_gpu.py tests/test_lora_gpu.py -m gpu -q -k "reference_forward or reference_generate or reference_model_fixture" --timeout 200 -p no:cacheprovider > $OUT/tests.log 2>&1; echo rc=$?; tail -25 $OUT/tests.log | cut -c1-250
