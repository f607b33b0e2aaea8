#!/bin/bash
# Round 2, call 16: attention backward with two-deep prefetch / 128-row blocks: parity tests, micro-probe, in-situ bench + kernel stats.
R=$PWD; OUT=$R/gpurun_out/r2c16; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=15 run tests_attn 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemma_gpu.py tests/test_f32_parity_gpu.py tests/test_lora_gpu.py tests/test_model_gpu.py tests/test_c2_width_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider
TAIL=4 run attn_probe 300 python tools/gpu_attn_bwd_probe.py
TAIL=1 run bench 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline
bash tools/kernel_stats.sh > $OUT/kernel_stats.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/stats2/s_kernel_stats.csv")))
for r in rows[:26]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(7), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), r["Percentage"].rjust(7))
PY
