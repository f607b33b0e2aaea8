#!/bin/bash
# round 4, call 10: HBM read bandwidth of weight-streaming access patterns (the decode GEMV's mapping and alternatives)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c10; mkdir -p $O
timeout 300 tools/probes/hbm_stream_probe 57344 8192 > $O/hbm_stream_gateup.txt 2>&1
cat $O/hbm_stream_gateup.txt
timeout 300 tools/probes/hbm_stream_probe 8192 28672 > $O/hbm_stream_down.txt 2>&1
cat $O/hbm_stream_down.txt
