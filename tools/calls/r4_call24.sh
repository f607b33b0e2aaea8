#!/bin/bash
# round 4, call 24 (call 23 again after the fix: compile-time tile index in the staged loop): tests, C4 at B = 8, 16, 32, 64 (B = 32 against the tiled kernels)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c24; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fused_rmsnorm or few_rows" > $O/pytest_kernels.txt 2>&1; tail -3 $O/pytest_kernels.txt
timeout 600 python -m pytest tests/test_generate_gpu.py tests/test_baseline_configs_gpu.py tests/test_gemma_gpu.py tests/test_qwen_gpu.py -q > $O/pytest_generate.txt 2>&1; tail -3 $O/pytest_generate.txt
for b in 8 16; do timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --batch $b > $O/bench_c4_b${b}.json 2>/dev/null; done
for b in 32 64; do timeout 900 python bench.py --workload c4 --steps 1 --warmup 1 --batch $b > $O/bench_c4_b${b}_stage.json 2>/dev/null; done
timeout 900 python bench.py --workload c4 --steps 1 --warmup 1 --batch 32 --opt 4=2 > $O/bench_c4_b32_tiled.json 2>/dev/null
for f in $O/bench*.json; do python - <<PY
import json
try:
    r=json.loads(open("$f").read().strip().splitlines()[-1])
    print("$f", "decode ms/token", round(r["decode_ms_per_token"],2), "frac", round(r["roofline"]["frac"],3), "tok/s", round(r["decode_tokens_per_sec"],1), "prefill ms", round(r["prefill_ms"],1))
except Exception as e: print("$f", "failed", e)
PY
done
