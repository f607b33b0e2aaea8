#!/bin/bash
# round 5, call 13: per-clip hip / torch distance ratios of the wav2vec2 tower at depth 24 (8 clips of 30 s; 8 clips of 10 s)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c13; mkdir -p $O
timeout 600 python tools/gpu_c5_tower_stage_probe.py 24 30 8 > $O/c5_tower_per_clip_30s.txt 2>&1; grep -v amdgpu.ids $O/c5_tower_per_clip_30s.txt | tail -3
timeout 600 python tools/gpu_c5_tower_stage_probe.py 24 10 8 > $O/c5_tower_per_clip_10s.txt 2>&1; grep -v amdgpu.ids $O/c5_tower_per_clip_10s.txt | tail -2
