#!/bin/bash
# round 5, call 8: the whole GPU suite (no -x) on the build of 00b165d + the near-tie form of the decode-batch token comparison;
# one C2 bench line for the box's speed
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c8; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2>$O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-400
