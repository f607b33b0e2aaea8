#!/bin/bash
# Round 2, call 15: stream-K GEMM variants: correctness / determinism check, then the cold-weight probe against the twins.
R=$PWD; OUT=$R/gpurun_out/r2c15; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=8 run sk_check_39 300 python tools/gpu_gemm_streamk_check.py 39
grep -c OK $OUT/sk_check_39.log; grep FAIL $OUT/sk_check_39.log | head -20
if grep -q ALL_OK $OUT/sk_check_39.log; then
  TAIL=8 run sk_check_rest 600 python tools/gpu_gemm_streamk_check.py 40,41,42
  grep FAIL $OUT/sk_check_rest.log | head -20
  TAIL=10 run sk_probe_llm 600 python tools/gpu_gemm_streamk_probe.py llm
  TAIL=10 run sk_probe_enc 600 python tools/gpu_gemm_streamk_probe.py enc
fi
