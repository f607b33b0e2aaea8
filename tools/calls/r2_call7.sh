#!/bin/bash
# Round 2, call 7: the merged-phase (PH = 2) eight-phase-kernel variants 31..34: correctness + race screen, cold-weight probe
# next to their four-phase twins (11, 16, 15/18, 17).
R=$PWD; OUT=$R/gpurun_out/r2c7; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
for v in 31 32 33 34; do TAIL=3 run check_v$v 120 python tools/gpu_gemm_check_variant.py $v; done
TAIL=9 run cold_probe 300 python tools/gpu_gemm_cold_probe.py 11,31,16,32,18,33,17,34
TAIL=9 run cold_probe_enc 300 python tools/gpu_gemm_cold_probe.py 11,31,16,32,18,33 enc
