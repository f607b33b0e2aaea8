#!/bin/bash
# Round 2, call 2: re-run the tests that failed in call 1 (flash-rounding bf16 oracle, 2-rank trainer), the issue-priority probe
# variants of the eight-phase GEMM (28 = none, 29 = load section prioritised, 30 = static second row) - correctness, then the
# cold-weight probe next to the production variant 11.
R=$PWD; OUT=$R/gpurun_out/r2c2; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=30 run tests_gpu 600 python -m pytest tests/test_bf16_rounding_points_gpu.py tests/test_dp_trainer_gpu.py tests/test_baseline_configs_gpu.py::test_c2_width_deeper_error_growth_is_bounded tests/test_checkpoint_gpu.py tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider
for v in 28 29 30; do TAIL=4 run check_v$v 120 python tools/gpu_gemm_check_variant.py $v; done
TAIL=10 run cold_probe 300 python tools/gpu_gemm_cold_probe.py 11,28,29,30
TAIL=10 run cold_probe_enc 300 python tools/gpu_gemm_cold_probe.py 11,16,28,29,30 enc
