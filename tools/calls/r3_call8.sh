#!/bin/bash
# round 3, call 8: one chain vs two chains on the scratch-free GEMM build, and the same with the previous GEMM object (whose kernels
# carried a private / scratch segment from round 2's residual-prefetch probe array) - same box, alternating
mkdir -p gpurun_out
run() {  # name, lib, extra args
  UVX_LIB=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $3 > gpurun_out/r3c8_bench_$1.json 2> gpurun_out/r3c8_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c8_bench_$1.json"))
r=d["roofline"]
print("%-26s ms/step %.2f loss %.5f gemm union %.2f TF/s %.1f" % ("$1", d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["achieved"]))
PY
}
NEW=ultravox_amd/libuvx.so
OLD=ultravox_amd/libuvx_oldgemm_probe.so
run new_chain1_a $NEW "--opt 11=0"
run new_chain2_a $NEW "--opt 11=2"
run old_chain1_a $OLD "--opt 11=0"
run old_chain2_a $OLD "--opt 11=2"
run new_chain1_b $NEW "--opt 11=0"
run new_chain2_b $NEW "--opt 11=2"
run old_chain1_b $OLD "--opt 11=0"
run old_chain2_b $OLD "--opt 11=2"
UVX_LIB=$NEW python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt 11=0 --gemm-table gpurun_out/r3c8_gemm_table_new_chain1.txt > /dev/null 2>&1
UVX_LIB=$OLD python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt 11=0 --gemm-table gpurun_out/r3c8_gemm_table_old_chain1.txt > /dev/null 2>&1
head -12 gpurun_out/r3c8_gemm_table_new_chain1.txt; head -12 gpurun_out/r3c8_gemm_table_old_chain1.txt
