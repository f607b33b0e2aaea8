#!/bin/bash
# round 5, call 7: whole GPU suite on the build with split-K in prefill / encoder inference / projector / decode batches > 16;
# c4 bench lines at B = 1, 8, 16, 32, 64; tiled split-K vs the weight-streaming kernels at 4 and 8 rows
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c7; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
timeout 300 python tools/gpu_gemm_splitk_probe.py 70b 4,8 34,0 > $O/splitk_probe_rows_4_8.txt 2>&1; grep -v amdgpu.ids $O/splitk_probe_rows_4_8.txt
for b in 1 8 16 32 64; do
  timeout 600 python bench.py --workload c4 --batch $b --steps 2 --warmup 1 > $O/bench_c4_b$b.json 2>$O/bench_c4_b$b.err
  tail -1 $O/bench_c4_b$b.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4 B=$b prefill_ms', round(r['prefill_ms'],2), 'decode ms/token', round(r['decode_ms_per_token'],2), 'tokens/s', round(r['decode_tokens_per_sec'],1), 'frac', round(r['roofline']['frac'],3))"
done
