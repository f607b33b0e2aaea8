#!/bin/bash
# round 6, call 27: uvx_llm_bwd_train_from (ABI 19) - the LLM backward from the first audio token on row-compacted gradients: bit-identity tests, the tests of
# the kernels it touches, then the C2 step with and without it (two repetitions, same box)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c27; mkdir -p $O
timeout 600 python -m pytest tests/test_prefix_skip_gpu.py -q -x 2>&1 | tail -15 | tee $O/pytest_new.txt
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_kl_gpu.py tests/test_dp_trainer_gpu.py -q 2>&1 | tail -6 | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4), 'gemm_ms', round(r['roofline']['gemm_ms_per_step'],2), 'mfu', round(r['mfu'],4))"; }
for rep in 1 2; do
for f in "full_backward:--no-prefix-skip" "from_first_audio_token:"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>$O/$name.err | tail -1 | line $name | tee -a $O/prefix_skip_ab.txt
done
done
