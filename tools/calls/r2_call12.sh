#!/bin/bash
# Round 2, call 12: m-major tile order for M > N GEMMs (option 7): kernel / model tests, same-box A/B, FETCH_SIZE pass per arm.
R=$PWD; OUT=$R/gpurun_out/r2c12; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=4 run tests_gpu 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_c2_width_gpu.py tests/test_baseline_configs_gpu.py tests/test_wav2vec2_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider
for arm in 1 0 1 0; do
  TAIL=1 run bench_mm$arm 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt 7=$arm --gemm-table $OUT/tab_mm$arm.txt
  grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' $OUT/bench_mm$arm.log | tr '\n' ' '; echo
done
grep "^   12000\|^    1504\|^    4096" $OUT/tab_mm1.txt; echo; grep "^   12000\|^    1504\|^    4096" $OUT/tab_mm0.txt
cd /tmp; export TMPDIR=/tmp
for arm in 1 0; do
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch$arm -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --opt 7=$arm > $OUT/rocprof_fetch$arm.log 2>&1
  python - <<PY
import csv, glob
n = s = 0
for f in glob.glob("$OUT/fetch$arm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt" in r["Kernel_Name"]:
            n += 1; s += float(r["Counter_Value"])
print("option 7 = $arm: FETCH_SIZE per GEMM launch (raw KB)", s / max(n, 1), "launches", n)
PY
  rm -rf $OUT/fetch$arm
done
