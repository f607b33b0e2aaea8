#!/bin/bash
# round 4, call 39: decode attention with the query-head group of a KV head dealt to several blocks (~256 blocks in flight): tests, C4 at B = 1, 8
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c39; mkdir -p $O
timeout 600 python -m pytest tests/test_generate_gpu.py tests/test_baseline_configs_gpu.py tests/test_gemma_gpu.py tests/test_qwen_gpu.py tests/test_gemma3_gpu.py -q > $O/pytest_generate.txt 2>&1; tail -2 $O/pytest_generate.txt
for b in 1 8; do timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --batch $b > $O/bench_c4_b${b}.json 2>/dev/null; done
for f in $O/bench*.json; do python - <<PY
import json
r=json.loads(open("$f").read().strip().splitlines()[-1])
print("$f", "decode ms/token", round(r["decode_ms_per_token"],2), "frac", round(r["roofline"]["frac"],3), "tok/s", round(r["decode_tokens_per_sec"],1))
PY
done
