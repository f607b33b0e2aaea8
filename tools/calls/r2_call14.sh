#!/bin/bash
# Round 2, call 14: top-layer supervised-row pair (uvx_llm_fwd_train / uvx_llm_bwd_train): parity tests, then same-box A/B.
R=$PWD; OUT=$R/gpurun_out/r2c14; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=30 run tests_gpu 1500 python -m pytest tests/test_model_gpu.py tests/test_gemma_gpu.py tests/test_wav2vec2_gpu.py tests/test_generate_gpu.py tests/test_bf16_rounding_points_gpu.py tests/test_c2_width_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -x
for i in 1 2; do
  UVX_TOP_LAYER_ROWS=0 TAIL=1 run bench_off_$i 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline
  UVX_TOP_LAYER_ROWS=1 TAIL=1 run bench_on_$i 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline
done
