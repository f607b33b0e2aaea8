#!/bin/bash
# round 5, call 33: split-K on the training step's deep-K single-round GEMMs (2528 x 4096 x 28672 / 14336 / 4096), every tile x factor, cold weights
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c33; mkdir -p $O
timeout 600 python tools/gpu_gemm_splitk_train_shapes_probe.py > $O/splitk_train_shapes.txt 2>&1; grep -v amdgpu.ids $O/splitk_train_shapes.txt
