#!/bin/bash
# Round 2, call 5: the wav2vec2 tower (C5) tests, Gemma tests again, the per-kernel bf16 bit-level test, a C5 bench smoke.
R=$PWD; OUT=$R/gpurun_out/r2c5; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=60 run tests_gpu 900 python -m pytest tests/test_wav2vec2_gpu.py tests/test_gemma_gpu.py tests/test_bf16_rounding_points_gpu.py::test_every_kernel_rounds_where_torch_bf16_rounds -m gpu -q --timeout 600 -p no:cacheprovider
TAIL=3 run bench_c5 400 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --gemm-table $OUT/gemm_table_c5.txt
head -40 $OUT/gemm_table_c5.txt
