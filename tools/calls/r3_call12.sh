#!/bin/bash
# round 3, call 12: the fused attention backward kernel - numerics (unit tests, model tests), then in-situ A/B against the kernel pair
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > gpurun_out/r3c12_tests_attn.log 2>&1
tail -15 gpurun_out/r3c12_tests_attn.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_c2_width_gpu.py tests/test_lora_gpu.py tests/test_kl_gpu.py tests/test_f32_parity_gpu.py -m gpu -q > gpurun_out/r3c12_tests_model.log 2>&1
tail -6 gpurun_out/r3c12_tests_model.log
run() {  # name, extra args
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c12_bench_$1.json 2> gpurun_out/r3c12_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c12_bench_$1.json"))
r=d["roofline"]
print("%-26s ms/step %.2f loss %.5f gemm union %.2f TF/s %.1f" % ("$1", d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["achieved"]))
PY
}
run fused_chain1_a "--opt 13=1,11=0"
run pair_chain1_a "--opt 13=0,11=0"
run fused_chain2_a "--opt 13=1,11=2"
run pair_chain2_a "--opt 13=0,11=2"
run fused_chain1_b "--opt 13=1,11=0"
run pair_chain1_b "--opt 13=0,11=0"
run fused_chain2_b "--opt 13=1,11=2"
run pair_chain2_b "--opt 13=0,11=2"
