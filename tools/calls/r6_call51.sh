#!/bin/bash
# round 6, call 51: the inference lines (BASELINE config 4) on the round's final build (two weight rows in flight in the one-row decode GEMV, grouped-query
# attention forward at >= 512 blocks): c4 at batch 1 / 8, c4s at batch 1
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c51; mkdir -p $O
for f in "c4_b1:--workload c4 --batch 1" "c4_b8:--workload c4 --batch 8" "c4s_b1:--workload c4s --batch 1"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 900 python bench.py $flags 2> $O/$name.err | tail -1 > $O/bench_$name.json
  python -c "
import json; r = json.load(open('$O/bench_$name.json')); print('$name', round(r['value'], 2), r['unit'], 'ttft ms', round(r.get('prefill_ms', 0), 2), 'decode ms/token', round(r['decode_ms_per_token'], 3), 'frac of 8 TB/s', round(r['roofline']['frac'], 4))" | tee -a $O/lines.txt
done
