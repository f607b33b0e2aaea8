#!/bin/bash
# round 5, call 12: the wav2vec2 tower stage probe at B = 1 x 30 s (the full-depth test's batch) and B = 4 x 30 s
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c12; mkdir -p $O
timeout 600 python tools/gpu_c5_tower_stage_probe.py 1,12,24 30 1 > $O/c5_tower_stage_probe_30s_b1.txt 2>&1; grep -v amdgpu.ids $O/c5_tower_stage_probe_30s_b1.txt | tail -5
timeout 600 python tools/gpu_c5_tower_stage_probe.py 24 30 4 > $O/c5_tower_stage_probe_30s_b4.txt 2>&1; grep -v amdgpu.ids $O/c5_tower_stage_probe_30s_b4.txt | tail -2
