#!/bin/bash
# round 5, call 29: the inference lines once more on the round's last build (after the register-resident norm): c4s B = 1, 8; c4 B = 1, 8, 16, 32
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c29; mkdir -p $O
timeout 300 python bench.py --workload c4s --steps 4 --warmup 2 > $O/bench_c4s_b1.json 2>/dev/null
timeout 300 python bench.py --workload c4s --batch 8 --steps 3 --warmup 2 > $O/bench_c4s_b8.json 2>/dev/null
for b in 1 8 16 32; do timeout 600 python bench.py --workload c4 --batch $b --steps 2 --warmup 1 > $O/bench_c4_b$b.json 2>/dev/null; done
for f in c4s_b1 c4s_b8 c4_b1 c4_b8 c4_b16 c4_b32; do tail -1 $O/bench_$f.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$f prefill_ms', round(r['prefill_ms'],2), 'decode ms/token', round(r['decode_ms_per_token'],2), 'tokens/s', round(r['decode_tokens_per_sec'],1), 'frac', round(r['roofline']['frac'],4))"; done
