#!/bin/bash
# Round 2, call 8: same-box A/B of the tile picker with the merged-phase kernels (option 6 = 1, the new default) against the
# round-1 four-phase set (6 = 0), then the GEMM / model / width tests on the new default.
R=$PWD; OUT=$R/gpurun_out/r2c8; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
for arm in 1 0 1 0; do
  TAIL=1 run bench_ph$arm 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt 6=$arm --gemm-table $OUT/tab_ph$arm.txt
  grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' $OUT/bench_ph$arm.log | tr '\n' ' '; echo
done
head -16 $OUT/tab_ph1.txt; head -16 $OUT/tab_ph0.txt
TAIL=6 run tests_gpu 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider
