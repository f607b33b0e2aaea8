#!/bin/bash
# round 6, call 33: the other workload lines on the ABI 19 build (backward from the first audio token on by default): c3 / c5 / q3 / g3 / l70 training steps
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c33; mkdir -p $O
for w in c3 c5 q3 g3 l70; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline 2> $O/bench_$w.err | tail -1 > $O/bench_$w.json
  python -c "
import json; r = json.loads(open('$O/bench_$w.json').read()); print('$w', round(r['ms_per_step'], 2), 'ms/step', round(r['value'], 1), r['unit'], 'mfu', round(r['mfu'], 4), 'gemm frac', round(r['roofline']['frac'], 4), 'from', r['config']['llm_backward_from_position'])" | tee -a $O/lines.txt
done
