#!/bin/bash
# round 6, call 47: the final build of the round (ABI 19: backward from the first audio token at any T / head_dim, grouped-query attention forward) - whole GPU suite, smoke, the no-flag bench line (what the driver runs)
# with its wall time, rocprofv3 kernel stats of the same workload, PMC traffic of the dominant GEMM family
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c47; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real | tee $O/bench_time.txt
python -c "
import json; r = json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print({k: r[k] for k in ('value','ms_per_step','mfu')}, r['roofline']['frac'], r['roofline']['traffic'], r['cpu_baseline']['value'], r['parity']['live']['logits_rel_l2_vs_f32_oracle'], r['config']['llm_backward_from_position'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o c2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-live-traffic > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
head -12 $O/kernel_stats.csv
