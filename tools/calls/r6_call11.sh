#!/bin/bash
# round 6, call 11: fixed-order RMSNorm weight-gradient sums (bit-reproducible training step): tests + the C2 line
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c11; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_lora_gpu.py tests/test_kernels_gpu.py tests/test_dp_trainer_gpu.py tests/test_kl_gpu.py -q 2>&1 | tail -5 | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4))"; }
for rep in 1 2; do
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | line ce | tee -a $O/ce.txt
done
