#!/bin/bash
# round 6, call 34: the grouped-query block form of the LLM's attention forward as the default (option 25 rule) - kernel tests, then the C2 step against option 25 = 5
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c34; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_generate_gpu.py -q 2>&1 | tail -4 | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'loss', round(r['loss'],4))"; }
for rep in 1 2; do
for f in "one_head_per_block:--opt 25=5" "grouped_query_blocks:"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>$O/$name.err | tail -1 | line $name | tee -a $O/attn_gq_step_ab.txt
done
done
