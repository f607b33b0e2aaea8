#!/bin/bash
# round 4, call 22: MFMA decode GEMV with the weights staged through a wave-private LDS tile (M = 3..16): tests, C4 at B = 4, 8, 16 against the straight fragment loads (option 4 = 2)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c22; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fused_rmsnorm or few_rows" > $O/pytest_kernels.txt 2>&1; tail -3 $O/pytest_kernels.txt
timeout 600 python -m pytest tests/test_generate_gpu.py tests/test_baseline_configs_gpu.py tests/test_gemma_gpu.py -q > $O/pytest_generate.txt 2>&1; tail -3 $O/pytest_generate.txt
for b in 4 8 16; do
  timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --batch $b > $O/bench_c4_b${b}_stage.json 2>/dev/null
  timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --batch $b --opt 4=2 > $O/bench_c4_b${b}_direct.json 2>/dev/null
done
for f in $O/bench*.json; do python - <<PY
import json
r=json.loads(open("$f").read().strip().splitlines()[-1])
print("$f", "decode ms/token", round(r["decode_ms_per_token"],2), "frac", round(r["roofline"]["frac"],3), "tok/s", round(r["decode_tokens_per_sec"],1))
PY
done
