#!/bin/bash
# Round 2, call 10: persistent merged-phase probe variants 35..38 (correctness, cold probe vs 31..34), and a same-box A/B of
# option 2 (SwiGLU backward fused into the down-projection dgrad epilogue) on the merged-phase kernels.
R=$PWD; OUT=$R/gpurun_out/r2c10; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
for v in 35 36 37 38; do TAIL=2 run check_v$v 120 python tools/gpu_gemm_check_variant.py $v; done
TAIL=9 run cold_probe_enc 300 python tools/gpu_gemm_cold_probe.py 31,35,32,36,33,37 enc
TAIL=9 run cold_probe 300 python tools/gpu_gemm_cold_probe.py 31,35,32,36,33,37
for arm in 0 1 0 1; do
  TAIL=1 run bench_sw$arm 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt 2=$arm
  grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' $OUT/bench_sw$arm.log | tr '\n' ' '; echo
done
