#!/bin/bash
# Round 2, call 6: same-box A/B of the residual prefetch in the GEMM epilogue (option 5), kernel + model tests on the new epilogue.
R=$PWD; OUT=$R/gpurun_out/r2c6; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=8 run tests_gpu 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_c2_width_gpu.py tests/test_bf16_rounding_points_gpu.py::test_every_kernel_rounds_where_torch_bf16_rounds -m gpu -q --timeout 600 -p no:cacheprovider
for arm in 1 0 1 0; do
  TAIL=1 run bench_pf$arm 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt 5=$arm --gemm-table $OUT/tab_pf$arm.txt
  grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' $OUT/bench_pf$arm.log | tr '\n' ' '; echo
done
paste <(awk 'NR>2{print $1,$2,$3,$8}' $OUT/tab_pf1.txt) <(awk 'NR>2{print $8}' $OUT/tab_pf0.txt) | head -14
