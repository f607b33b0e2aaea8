#!/bin/bash
# round 3, call 11: grid caps for the HBM-bound kernels between the GEMMs (option 5: SwiGLU backward, RMSNorm forward / backward) inside the
# two-chain schedule; log-mel tests with the tightened bounds; the schedule tuner's GPU test
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_dp_trainer_gpu.py tests/test_model_gpu.py -m gpu -q -k "logmel or autotuner or rmsnorm or swiglu or train_step" > gpurun_out/r3c11_tests.log 2>&1
tail -5 gpurun_out/r3c11_tests.log
run() {  # name, extra args
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c11_bench_$1.json 2> gpurun_out/r3c11_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c11_bench_$1.json"))
r=d["roofline"]
print("%-26s ms/step %.2f loss %.5f gemm union %.2f TF/s %.1f" % ("$1", d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["achieved"]))
PY
}
run cap0_chain2_a "--opt 5=0,11=2"
run cap64_chain2 "--opt 5=64,11=2"
run cap128_chain2 "--opt 5=128,11=2"
run cap256_chain2 "--opt 5=256,11=2"
run cap512_chain2 "--opt 5=512,11=2"
run cap1024_chain2 "--opt 5=1024,11=2"
run cap0_chain2_b "--opt 5=0,11=2"
run cap0_chain1 "--opt 5=0,11=0"
run cap512_chain1 "--opt 5=512,11=0"
