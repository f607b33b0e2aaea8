#!/bin/bash
# round 6, call 37: the decode step's row-streaming GEMV at B = 1 with 8 / 16 weight rows in flight per wave instead of 4 (tuning option 26): 8B and 70B, ms per token
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c37; mkdir -p $O
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 decode ms/token', round(r['decode_ms_per_token'],3), 'frac', round(r['roofline']['frac'],4), 'prefill ms', round(r.get('prefill_ms', 0),2))"; }
for rep in 1 2; do
for wl in c4s c4; do
for f in "rows4:" "rows2:--opt 26=2"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 600 python bench.py --workload $wl --batch 1 --steps 3 --warmup 1 $flags 2>$O/$wl.$name.err | tail -1 | line "$wl $name" | tee -a $O/gemv_rows_ab.txt
done
done
done
