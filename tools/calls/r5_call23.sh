#!/bin/bash
# round 5, call 23: RMSNorm forward with the row in registers (option 18 = 0, the default) against the two-pass kernel (18 = 1): kernel tests, then
# same-box A/B on the 70B decode step at B = 8 / 16, the 8B model at B = 8, and the C2 training step
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c23; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_rounding_points_gpu.py tests/test_generate_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest.txt
for b in 8 16; do for o in 0 1 0 1; do
  timeout 600 python bench.py --workload c4 --batch $b --steps 2 --warmup 1 --opt 18=$o 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4 B=$b option18=$o prefill_ms', round(r['prefill_ms'],2), 'decode ms/token', round(r['decode_ms_per_token'],3), 'frac', round(r['roofline']['frac'],4))" | tee -a $O/norm_reg_ab.txt
done; done
for o in 0 1; do
  timeout 300 python bench.py --workload c4s --batch 8 --steps 3 --warmup 2 --opt 18=$o 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4s B=8 option18=$o prefill_ms', round(r['prefill_ms'],2), 'decode ms/token', round(r['decode_ms_per_token'],3))" | tee -a $O/norm_reg_ab.txt
done
for o in 0 1 0 1; do
  timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --opt 18=$o 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c2 option18=$o ms/step', round(r['ms_per_step'],3))" | tee -a $O/norm_reg_ab.txt
done
