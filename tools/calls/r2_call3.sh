#!/bin/bash
# Round 2, call 3: whole -m gpu suite on the new GEMM default (static issue priority) + the per-kernel bf16 bit-level tests,
# PMC passes on the production GEMM, the whole-step CPU baseline, a bench run.
R=$PWD; OUT=$R/gpurun_out/r2c3; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=40 run tests_gpu 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider
TAIL=40 run pmc 400 bash tools/pmc_gemm_prod.sh
TAIL=3 run cpu_full 400 python bench.py --cpu-baseline-full $OUT/cpu_baseline_full.json
TAIL=3 run bench 300 python bench.py --steps 10 --warmup 3 --gemm-table $OUT/gemm_table.txt
head -24 $OUT/gemm_table.txt
