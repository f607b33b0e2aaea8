#!/bin/bash
# round 5, call 6: the decode step's batches (M = 16 ... 128 rows) through the TILED kernels with split-K against the weight-streaming
# skinny kernels ("auto" for M <= 64): can a 128 x 256 tile cut 8-16 ways stream the weights faster than the staged MFMA kernel?
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c6; mkdir -p $O
timeout 600 python tools/gpu_gemm_splitk_probe.py 70b 16,32,64,128 34,60,0 > $O/splitk_probe_decode_rows.txt 2>&1; grep -v amdgpu.ids $O/splitk_probe_decode_rows.txt
