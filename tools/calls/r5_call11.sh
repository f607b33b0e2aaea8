#!/bin/bash
# round 5, call 11: the wav2vec2 tower stage probe at the full-depth test's clip length (30 s)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c11; mkdir -p $O
timeout 600 python tools/gpu_c5_tower_stage_probe.py 1,6,12,24 30 > $O/c5_tower_stage_probe_30s.txt 2>&1; grep -v amdgpu.ids $O/c5_tower_stage_probe_30s.txt | tail -8
timeout 600 python tools/gpu_c5_tower_stage_probe.py 24 20 > $O/c5_tower_stage_probe_20s.txt 2>&1; grep -v amdgpu.ids $O/c5_tower_stage_probe_20s.txt | tail -2
