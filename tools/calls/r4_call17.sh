#!/bin/bash
# round 4, call 17: encoder forward as two half-batch chains on two streams (probe)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c17; mkdir -p $O
timeout 300 python tools/gpu_encoder_two_stream_probe.py > $O/enc_two_stream.txt 2>&1; grep -v amdgpu.ids $O/enc_two_stream.txt | tail -6
