#!/bin/bash
# round 5, call 26: the whole GPU suite, smoke and the no-flag bench line on the last commit of the round
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c26; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
T0=$(date +%s.%N); timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; T1=$(date +%s.%N); echo "wall seconds of python bench.py: $(python -c "print(round($T1 - $T0, 1))")"
tail -1 $O/bench_default.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'frac', round(r['roofline']['frac'],4), 'traffic', round(r['roofline']['traffic']/1e6,1), 'MB', r['roofline']['traffic_source'][:40])"
