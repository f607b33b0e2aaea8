#!/bin/bash
# round 6, call 19: beam search (generate(num_beams > 1)) GPU tests + the option-24 tests after the default flip
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c19; mkdir -p $O
timeout 1200 python -m pytest tests/test_generate_gpu.py tests/test_kernels_gpu.py -q -k "beam or rmsnorm or decode" 2>&1 | tail -25 | tee $O/pytest.txt
