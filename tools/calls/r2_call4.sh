#!/bin/bash
# Round 2, call 4: the Gemma backbone (C5) tests, the per-kernel bf16 bit-level test (fixed ulp metric), the uvx_comm_* route.
R=$PWD; OUT=$R/gpurun_out/r2c4; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=60 run tests_gpu 900 python -m pytest tests/test_gemma_gpu.py tests/test_bf16_rounding_points_gpu.py::test_every_kernel_rounds_where_torch_bf16_rounds tests/test_dp_trainer_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider
