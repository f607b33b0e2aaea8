#!/bin/bash
# Round 2, first GPU call: the whole -m gpu suite (incl. the new C3 / C4-width / deeper-C2 / bf16-oracle / 2-rank trainer tests),
# the eight-phase GEMM timeline probes, and a bench run with the per-shape GEMM table.
R=$PWD; OUT=$R/gpurun_out/r2c1; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=40 run tests_gpu 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --durations=15
TAIL=14 run gemm_timeline   120 python tools/gpu_gemm_timeline.py
TAIL=14 run gemm_timeline_k 120 python tools/gpu_gemm_timeline.py 2528 4096 4096
TAIL=3 run bench 300 python bench.py --steps 10 --warmup 3 --gemm-table $OUT/gemm_table.txt
cat $OUT/gemm_table.txt | head -40
