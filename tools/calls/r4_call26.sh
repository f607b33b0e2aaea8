#!/bin/bash
# round 4, call 26: staged GEMV with two activation chunks in registers at MT >= 2 (two blocks per CU at MT = 2): tests, C4 at B = 32, 64
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c26; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fused_rmsnorm or few_rows" > $O/pytest_kernels.txt 2>&1; tail -2 $O/pytest_kernels.txt
timeout 600 python -m pytest tests/test_baseline_configs_gpu.py tests/test_generate_gpu.py -q > $O/pytest_generate.txt 2>&1; tail -2 $O/pytest_generate.txt
for b in 24 32 64; do timeout 900 python bench.py --workload c4 --steps 1 --warmup 1 --batch $b > $O/bench_c4_b${b}.json 2>/dev/null; done
for f in $O/bench*.json; do python - <<PY
import json
try:
    r=json.loads(open("$f").read().strip().splitlines()[-1])
    print("$f", "decode ms/token", round(r["decode_ms_per_token"],2), "frac", round(r["roofline"]["frac"],3), "tok/s", round(r["decode_tokens_per_sec"],1), "prefill ms", round(r["prefill_ms"],1))
except Exception as e: print("$f", "failed", e)
PY
done
