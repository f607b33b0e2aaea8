#!/bin/bash
# round 5, call 24: the default bench line with roofline.traffic measured in the run (two rocprofv3 --pmc sub-runs), wall time of the whole command
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c24; mkdir -p $O
T0=$(date +%s.%N); python bench.py > $O/bench_default.json 2>$O/bench_default.err; T1=$(date +%s.%N); echo "wall seconds of python bench.py: $(python -c "print(round($T1 - $T0, 1))")"
tail -1 $O/bench_default.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms/step', round(r['ms_per_step'],2), 'frac', round(r['roofline']['frac'],4)); print('traffic', r['roofline']['traffic'], '|', r['roofline']['traffic_source']); print(r['roofline']['traffic_detail']); print('cpu', r['cpu_baseline']['value'], 'parity', r['parity']['logits_rel_l2_vs_f32_oracle'])"
tail -3 $O/bench_default.err | cut -c1-300
