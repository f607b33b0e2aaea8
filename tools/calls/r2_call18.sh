#!/bin/bash
# Round 2, call 18: full -m gpu suite after the attention changes (longest-first forward order included) + attention probe + bench.
R=$PWD; OUT=$R/gpurun_out/r2c18; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=6 run tests_gpu 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider
TAIL=3 run attn_probe 300 python tools/gpu_attn_bwd_probe.py
TAIL=1 run bench 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline
