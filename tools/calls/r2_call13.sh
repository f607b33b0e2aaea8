#!/bin/bash
# Round 2, call 13: generate() on the Gemma backbone; generate tests for Llama unchanged.
R=$PWD; OUT=$R/gpurun_out/r2c13; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=30 run tests_gpu 900 python -m pytest tests/test_gemma_gpu.py tests/test_generate_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider
