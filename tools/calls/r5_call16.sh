#!/bin/bash
# round 5, call 16 (ran on commit a7a5264, since reverted - profiles/r05_decode_last_block_norm_negative.txt): the RMSNorm that follows a 3..16-row linear, by the block that finishes last (option 17): tests, then same-box A/B of the
# 70B decode step at B = 4, 8, 16 and of the 8B model at B = 8
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c16; mkdir -p $O
timeout 900 python -m pytest tests/test_generate_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest.txt
for b in 8 4 16; do for o in 1 0; do
  timeout 600 python bench.py --workload c4 --batch $b --steps 2 --warmup 1 --opt 17=$o 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4 B=$b option17=$o prefill_ms', round(r['prefill_ms'],2), 'decode ms/token', round(r['decode_ms_per_token'],3), 'frac', round(r['roofline']['frac'],4))" | tee -a $O/last_block_norm_ab.txt
done; done
for o in 1 0; do
  timeout 300 python bench.py --workload c4s --batch 8 --steps 3 --warmup 2 --opt 17=$o 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4s B=8 option17=$o prefill_ms', round(r['prefill_ms'],2), 'decode ms/token', round(r['decode_ms_per_token'],3))" | tee -a $O/last_block_norm_ab.txt
done
