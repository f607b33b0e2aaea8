#!/bin/bash
# round 3, call 48: roofline events on every 4th step (default) vs every step vs none, same box
run() { timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('%-10s ms/step %.3f  gemm ms/step %s  TF/s %s  launches/step %s  %s' % ('$1', d['ms_per_step'], r.get('gemm_ms_per_step'), r.get('achieved'), r.get('launches_per_step'), r.get('profiled_steps')))"; }
run warm ""; run every4 ""; run every1 "--prof-every 1"; run noprof "--no-prof"; run every4_b ""; run every1_b "--prof-every 1"; run noprof_b "--no-prof"
timeout 900 python -m pytest tests/test_dp_trainer_gpu.py tests/test_smoke_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -iE "passed|failed|error" | tail -3
