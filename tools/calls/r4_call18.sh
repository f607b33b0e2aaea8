#!/bin/bash
# round 4, call 18: tile A/B on the encoder's fc1 shape (12000 x 4096 x 1024, bias + GELU epilogue): 256-row (picked) / 192-row / 160-row tiles, in situ
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c18; mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
timeout 200 $B --gemm-table $O/table_A.txt > $O/bench_A.json 2>/dev/null
timeout 200 $B --gemm-override 12000x4096x1024=32 --gemm-table $O/table_v32.txt > $O/bench_v32.json 2>/dev/null
timeout 200 $B --gemm-override 12000x4096x1024=33 --gemm-table $O/table_v33.txt > $O/bench_v33.json 2>/dev/null
timeout 200 $B --gemm-override 12000x1024x4096=31 --gemm-table $O/table_fc2_v31.txt > $O/bench_fc2_v31.json 2>/dev/null
timeout 200 $B --gemm-table $O/table_A2.txt > $O/bench_A2.json 2>/dev/null
for f in A v32 v33 fc2_v31 A2; do python - <<PY
import json
r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f", "ms/step", round(r["ms_per_step"],2), "gemm ms", round(r["roofline"]["gemm_ms_per_step"],2), "TF/s", round(r["roofline"]["achieved"],1))
PY
done
grep -h "12000 *4096 *1024\|12000 *1024 *4096" $O/table_A.txt $O/table_v32.txt $O/table_v33.txt $O/table_fc2_v31.txt $O/table_A2.txt
