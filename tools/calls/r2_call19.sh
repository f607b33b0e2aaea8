#!/bin/bash
# Round 2, call 19: log-mel DFT on the f32 matrix cores: frontend / model tests, then kernel stats of a bench run.
R=$PWD; OUT=$R/gpurun_out/r2c19; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=6 run tests 400 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_f32_parity_gpu.py -m gpu -q -x --timeout 200 -p no:cacheprovider
grep -q passed $OUT/tests.log || exit 1
bash tools/kernel_stats.sh > $OUT/kernel_stats.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/stats2/s_kernel_stats.csv")))
for r in rows[:40]:
    if "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
    print(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60].ljust(60), r["Calls"].rjust(7), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), r["Percentage"].rjust(7))
PY
tail -1 $R/gpurun_out/stats2/log.txt | cut -c1-200
