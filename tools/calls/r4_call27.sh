#!/bin/bash
# round 4, call 27: in-situ tile A/B on the LLM shapes with partly filled rounds / heavy epilogues (bench.py --gemm-override), same box
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c27; mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
run() { timeout 200 $B $2 --gemm-table $O/table_$1.txt > $O/bench_$1.json 2>/dev/null; python - <<PY
import json
r=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("$1".ljust(22), "ms/step", round(r["ms_per_step"],2), "gemm ms", round(r["roofline"]["gemm_ms_per_step"],2), "TF/s", round(r["roofline"]["achieved"],1))
PY
grep -h "2528 *28672 *4096\|2528 *14336 *4096\|2528 *6144 *4096" $O/table_$1.txt; }
run A ""
run gu33 "--gemm-override 2528x28672x4096=33"
run gu32 "--gemm-override 2528x28672x4096=32"
run dn33 "--gemm-override 2528x14336x4096=33"
run dn32 "--gemm-override 2528x14336x4096=32"
run dn34 "--gemm-override 2528x14336x4096=34"
run qkv33 "--gemm-override 2528x6144x4096=33"
run A2 ""
