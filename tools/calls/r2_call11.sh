#!/bin/bash
# Round 2, call 11: persistent merged-phase probe variants 35..37 vs 31..33 (cold probe), attention tests after folding delta
# into the dQ kernel, bench.
R=$PWD; OUT=$R/gpurun_out/r2c11; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=5 run tests_gpu 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_lora_gpu.py tests/test_gemma_gpu.py tests/test_c2_width_gpu.py tests/test_f32_parity_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider
TAIL=9 run cold_probe_enc 300 python tools/gpu_gemm_cold_probe.py 31,35,32,36,33,37 enc
TAIL=9 run cold_probe 300 python tools/gpu_gemm_cold_probe.py 31,35,32,36,33,37
TAIL=1 run bench 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' $OUT/bench.log | tr '\n' ' '; echo
