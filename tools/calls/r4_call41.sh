# round 4, call 41: the full GPU suite on the round's final build (after the decode-attention head split and the header edit).
# 258 passed in 216 s.
export PYTHONPATH=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c41
timeout 1400 python -m pytest tests -x -q -m gpu -rs > gpurun_out/r4c41/pytest_gpu.txt 2>&1
grep -E "passed|failed|skipped" gpurun_out/r4c41/pytest_gpu.txt | tail -3
