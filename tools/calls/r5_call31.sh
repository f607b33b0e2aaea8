#!/bin/bash
# round 5, call 31: the decode step's linears (M = 1, 8, 32 rows x the 70B shapes) on cold weights against hipBLASLt (torch.matmul), same probe as call 30
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c31; mkdir -p $O
timeout 900 python tools/gpu_prefill_vs_blaslt_probe.py 70b 1,8,32 > $O/decode_vs_blaslt.txt 2>&1; grep -v "amdgpu.ids" $O/decode_vs_blaslt.txt
