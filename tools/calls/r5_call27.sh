#!/bin/bash
# round 5, call 27: does the two-chain LLM schedule / the schedule tuner help on whatever box this call lands on?  (call 26's box read 85.3 ms per step,
# call 21's 76.7 with the same kernels.)  default, --opt 11=2,13=0 (two chains + attention backward pair), --opt 11=2 (two chains, fused attention
# backward), --autotune; each twice, interleaved
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c27; mkdir -p $O
for rep in 1 2; do
  for f in "default:" "two_chains_pair:--opt 11=2,13=0" "two_chains_fused:--opt 11=2" "autotune:--autotune"; do
    name=${f%%:*}; flags=${f#*:}
    timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline $flags 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$name ms/step', round(r['ms_per_step'],2), 'gemm frac', round(r['roofline']['frac'],4), 'schedule', r['llm_schedule']['chosen'], r['llm_schedule']['trial_ms'])" | tee -a $O/schedule_ab.txt
  done
done
