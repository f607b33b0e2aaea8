#!/bin/bash
# round 3, call 13: 64-query steps in the fused attention backward (option 13 = 2), query tiles per wave of the encoder attention (qt 2 / 3 / 4
# on the transposing-read kernels), one chain
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "fused_attention" > gpurun_out/r3c13_tests.log 2>&1; tail -3 gpurun_out/r3c13_tests.log
run() {  # name, extra args
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c13_bench_$1.json 2> gpurun_out/r3c13_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c13_bench_$1.json"))
r=d["roofline"]
print("%-26s ms/step %.2f loss %.5f gemm union %.2f TF/s %.1f" % ("$1", d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["achieved"]))
PY
}
run base_a ""
run step64_a "--opt 13=2"
run qt3_a "--attn-qt 3"
run qt4_a "--attn-qt 4"
run base_b ""
run step64_b "--opt 13=2"
run qt3_b "--attn-qt 3"
run qt1 "--attn-qt 1"
