#!/bin/bash
# round 5, call 22: the whole GPU suite, smoke and the default bench line on the build with ABI 15 (plain projector activations, the wav2vec2 layer-norm
# family, the batched LoRA reduces) - the suite twice, to see a flaky test if there is one
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c22; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_1.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest_gpu_1.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-420
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu_2.txt 2>&1; grep "passed\|failed\|^FAILED" $O/pytest_gpu_2.txt
