#!/bin/bash
# round 5, call 14: RMSNorm inside the split-K reduce (option 17): tests, then same-box A/B on the 8B prefill (c4s), the 70B prefill (B = 1) and the
# 70B decode step at B = 32 / 64 (tiled split-K path); the wav2vec2-large multi-clip calibration test
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c14; mkdir -p $O
timeout 900 python -m pytest tests/test_generate_gpu.py tests/test_wav2vec2_gpu.py tests/test_gemma_gpu.py tests/test_qwen_gpu.py tests/test_gemma3_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest.txt
for o in 1 0 1 0; do
  timeout 300 python bench.py --workload c4s --steps 4 --warmup 2 --opt 17=$o 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4s option17=$o prefill_ms', round(r['prefill_ms'],3), 'decode ms/token', round(r['decode_ms_per_token'],3))" | tee -a $O/reduce_norm_ab.txt
done
for b in 1 32 64; do for o in 1 0; do
  timeout 600 python bench.py --workload c4 --batch $b --steps 2 --warmup 1 --opt 17=$o 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4 B=$b option17=$o prefill_ms', round(r['prefill_ms'],2), 'decode ms/token', round(r['decode_ms_per_token'],3), 'frac', round(r['roofline']['frac'],4))" | tee -a $O/reduce_norm_ab.txt
done; done
