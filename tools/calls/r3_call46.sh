#!/bin/bash
# round 3, call 46: what do the per-GEMM HIP events of bench.py's roofline measurement cost?  (--no-prof: no events)
run() { timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline $2 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-10s ms/step %.3f' % ('$1', d['ms_per_step']))"; }
run warm ""; run prof ""; run noprof "--no-prof"; run prof_b ""; run noprof_b "--no-prof"; run prof_c ""; run noprof_c "--no-prof"
