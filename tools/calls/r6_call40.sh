#!/bin/bash
# round 6, call 40: splitk_reduce_norm_wide_k (one chunk per thread, the narrow kernel's summation order) - bit-identity tests, then time-to-first-token of the
# 70B and 8B models with it (default) and with option 28 = 1 (the narrow kernel)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c40; mkdir -p $O
timeout 900 python -m pytest tests/test_generate_gpu.py -q -x -k "splitk or reduce" 2>&1 | tail -4 | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 prefill ms', round(r.get('prefill_ms', 0),3), 'decode ms/token', round(r['decode_ms_per_token'],3))"; }
for rep in 1 2; do
for wl in c4s c4; do
for f in "narrow:--opt 28=1" "wide:"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 600 python bench.py --workload $wl --batch 1 --steps 3 --warmup 1 $flags 2>$O/$wl.$name.err | tail -1 | line "$wl $name" | tee -a $O/reduce_norm_wide_ab.txt
done
done
done
