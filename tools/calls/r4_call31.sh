#!/bin/bash
# round 4, call 31: software pipelining of the frozen encoder (log-mel + encoder of step i + 1 on a side stream during step i): same-box A/B, identical loss expected
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c31; mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
run() { timeout 200 $B $2 > $O/bench_$1.json 2> $O/err_$1.txt; python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    print("$1".ljust(14), "ms/step", round(r["ms_per_step"],2), "value", round(r["value"],1), "loss", r["loss"], "gemm ms", round(r["roofline"]["gemm_ms_per_step"],2))
except Exception as e: print("$1 failed", e); print(open("$O/err_$1.txt").read()[-1500:])
PY
}
run a1 ""
run pf1 "--prefetch-encoder"
run a2 ""
run pf2 "--prefetch-encoder"
run kl "--loss kl"
run kl_pf "--loss kl --prefetch-encoder"
