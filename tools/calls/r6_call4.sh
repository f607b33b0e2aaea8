#!/bin/bash
# round 6, call 4: forward attention with the V fragments preloaded before the softmax (option 20 = 2), the default bench line with its LIVE parity + traffic objects,
# and the stale bench lines (c3, c5, q3, g3, l70: round-3 records) refreshed on this build
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c4; mkdir -p $O
timeout 300 python tools/gpu_attn_r6_probe.py 3 2>&1 | grep -v amdgpu.ids | grep "fwd" | tee $O/attn_fwd_probe.txt
( time timeout 900 python bench.py ) > $O/bench_c2.json 2> $O/bench_c2.err; tail -3 $O/bench_c2.err; python -c "
import json; r = json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print({k: r[k] for k in ('value','ms_per_step','mfu')}); print(json.dumps(r.get('parity',{}).get('live'), indent=1)); print(r['roofline']['frac'], r['roofline']['traffic'], r['cpu_baseline'].get('value'))"
for w in c3 c5 q3 g3 l70; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline 2> $O/bench_$w.err | tail -1 > $O/bench_$w.json
  python -c "
import json; r = json.loads(open('$O/bench_$w.json').read()); print('$w', {k: r.get(k) for k in ('value','ms_per_step','mfu')}, (r.get('roofline') or {}).get('frac'))" || tail -5 $O/bench_$w.err
done
