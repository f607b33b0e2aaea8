#!/bin/bash
# round 4, call 21: halving-butterfly lane reduction in the row-streaming GEMV + 4 / 8 activation rows (option 4 = 3) against the MFMA mapping: tests, C4 at B = 1, 2, 4, 8
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c21; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fused_rmsnorm or few_rows" > $O/pytest_kernels.txt 2>&1; tail -3 $O/pytest_kernels.txt
timeout 600 python -m pytest tests/test_generate_gpu.py tests/test_baseline_configs_gpu.py -q > $O/pytest_generate.txt 2>&1; tail -3 $O/pytest_generate.txt
for b in 1 2; do timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --batch $b > $O/bench_c4_b${b}.json 2>/dev/null; done
for b in 4 8; do
  timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --batch $b > $O/bench_c4_b${b}_default.json 2>/dev/null
  timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --batch $b --opt 4=3 > $O/bench_c4_b${b}_rows.json 2>/dev/null
done
for f in $O/bench*.json; do python - <<PY
import json
r=json.loads(open("$f").read().strip().splitlines()[-1])
print("$f", "decode ms/token", round(r["decode_ms_per_token"],2), "frac", round(r["roofline"]["frac"],3))
PY
done
