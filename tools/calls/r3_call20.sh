#!/bin/bash
# round 3, call 20: fused attention backward with its two steps per chunk unrolled (232 VGPRs) vs rolled (194) - alternate builds, same box
mkdir -p gpurun_out
run() {  # name, lib
  UVX_LIB=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c20_bench_$1.json 2> gpurun_out/r3c20_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c20_bench_$1.json"))
print("%-16s ms/step %.2f loss %.5f" % ("$1", d["ms_per_step"], d["loss"]))
PY
}
UVX_LIB=ultravox_amd/libuvx_unroll_probe.so python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "fused_attention" 2>&1 | tail -2
run rolled_a ultravox_amd/libuvx.so
run unrolled_a ultravox_amd/libuvx_unroll_probe.so
run rolled_b ultravox_amd/libuvx.so
run unrolled_b ultravox_amd/libuvx_unroll_probe.so
