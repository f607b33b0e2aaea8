#!/bin/bash
# round 3, call 5: N-chain schedule (option 11 = 2 / 3 / 4), inverse RoPE fused into the attention backward (option 14), 64-row steps in
# the attention backward (option 13), forward(past_key_values); A/B inside bench.py on one box
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py tests/test_generate_gpu.py tests/test_kernels_gpu.py tests/test_lora_gpu.py tests/test_gemma_gpu.py -m gpu -q > gpurun_out/r3c5_tests.log 2>&1
tail -6 gpurun_out/r3c5_tests.log
run() {  # name, extra args
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c5_bench_$1.json 2> gpurun_out/r3c5_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c5_bench_$1.json"))
r=d["roofline"]
print("%-22s ms/step %.2f loss %.5f gemm union %.2f TF/s %.1f" % ("$1", d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["achieved"]))
PY
}
run chains2_a "--opt 11=2"
run chains4_a "--opt 11=4"
run chains3_a "--opt 11=3"
run rope_separate "--opt 14=0"
run st64 "--opt 13=1"
run chains2_b "--opt 11=2"
run chains4_b "--opt 11=4"
run rope_separate_b "--opt 14=0"
run st64_b "--opt 13=1"
