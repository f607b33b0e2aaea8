#!/bin/bash
# round 4, call 15: which GEMV serves B = 2 / 3 / 4 - row-streaming (default) against the MFMA mapping (option 4 = 2), 70B decode
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c15; mkdir -p $O
for b in 2 3 4; do
  timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --batch $b > $O/bench_c4_b${b}_rows.json 2>/dev/null
  timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --batch $b --opt 4=2 > $O/bench_c4_b${b}_mfma.json 2>/dev/null
done
for f in $O/*.json; do python - <<PY
import json
r=json.loads(open("$f").read().strip().splitlines()[-1])
print("$f", "decode ms/token", round(r["decode_ms_per_token"],2), "frac", round(r["roofline"]["frac"],3))
PY
done
