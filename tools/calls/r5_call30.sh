#!/bin/bash
# round 5, call 30: the prefill GEMMs on cold weights against hipBLASLt (torch.matmul) as a second opinion, M = 316 / 632, 8B and 70B linears
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c30; mkdir -p $O
timeout 900 python tools/gpu_prefill_vs_blaslt_probe.py all 316,632 > $O/prefill_vs_blaslt.txt 2>&1; grep -v "amdgpu.ids" $O/prefill_vs_blaslt.txt
