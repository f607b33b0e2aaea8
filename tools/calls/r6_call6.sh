#!/bin/bash
# round 6, call 6: which epilogue breaks bit-identity at rank 4 (test of call 5) - verbose rerun
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c6; mkdir -p $O
timeout 600 python -m pytest tests/test_lora_gpu.py -q -k "epilogues" 2>&1 | grep -v "^$" | tail -40 | tee $O/pytest_lora_epilogues.txt
timeout 600 python -m pytest tests/test_lora_gpu.py -q -k "llm_only" 2>&1 | tail -30 | tee $O/pytest_llm_only.txt
