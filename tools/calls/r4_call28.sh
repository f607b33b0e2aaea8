#!/bin/bash
# round 4, call 28: 2528 x 14336 x 4096 on the 160-row tile (3.5 rounds) against the 256-row tile (2.19 rounds), in situ, alternating, another box
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c28; mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
run() { timeout 200 $B $2 --gemm-table $O/table_$1.txt > $O/bench_$1.json 2>/dev/null; python - <<PY
import json
r=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("$1".ljust(10), "ms/step", round(r["ms_per_step"],2), "gemm ms", round(r["roofline"]["gemm_ms_per_step"],2), "TF/s", round(r["roofline"]["achieved"],1))
PY
grep -h "2528 *14336 *4096\|1504 *4096 *8192" $O/table_$1.txt; }
run A1 ""
run dn33a "--gemm-override 2528x14336x4096=33,1504x4096x8192=34"
run A2 ""
run dn33b "--gemm-override 2528x14336x4096=33,1504x4096x8192=34"
