#!/bin/bash
# round 3, call 47: roofline events attached to the GEMM dispatches (hipExtLaunchKernelGGL) instead of recorded around them: overhead and sanity
run() { timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline $2 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('%-10s ms/step %.3f  gemm ms/step %s  TF/s %s  launches/step %s' % ('$1', d['ms_per_step'], r.get('gemm_ms_per_step'), r.get('achieved'), r.get('launches_per_step')))"; }
run warm ""; run prof ""; run noprof "--no-prof"; run prof_b ""; run noprof_b "--no-prof"
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --gemm-table gpurun_out/r3c47_table.txt > /dev/null 2>&1; head -8 gpurun_out/r3c47_table.txt
timeout 600 python -m pytest tests/test_bench_gpu.py tests/test_dp_trainer_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -3
