#!/bin/bash
# round 3, call 10: fused RMSNorm after batching its partial-sum loads - in-situ A/B (x one / two chains)
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py -m gpu -q -k "rmsnorm_fused or train_step" > gpurun_out/r3c10_tests.log 2>&1
tail -5 gpurun_out/r3c10_tests.log
run() {  # name, extra args
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c10_bench_$1.json 2> gpurun_out/r3c10_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c10_bench_$1.json"))
r=d["roofline"]
print("%-26s ms/step %.2f loss %.5f gemm union %.2f TF/s %.1f" % ("$1", d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["achieved"]))
PY
}
run norm1_chain2_a "--opt 13=1,11=2"
run norm0_chain2_a "--opt 13=0,11=2"
run norm1_chain1_a "--opt 13=1,11=0"
run norm0_chain1_a "--opt 13=0,11=0"
run norm1_chain2_b "--opt 13=1,11=2"
run norm0_chain2_b "--opt 13=0,11=2"
run norm1_chain1_b "--opt 13=1,11=0"
run norm0_chain1_b "--opt 13=0,11=0"
