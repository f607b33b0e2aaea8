#!/bin/bash
# round 3, call 7: SwiGLU-backward epilogue through the LDS stage (option 2 = 2) - bit-exactness test + in-situ A/B against the separate
# kernel (0) and the fragment-layout epilogue (1); mask-holes test; the scratch-free GEMM build is in every arm
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_c2_width_gpu.py -m gpu -q -k "swiglu or holes or gemm or c2_width or train_step" > gpurun_out/r3c7_tests.log 2>&1
tail -6 gpurun_out/r3c7_tests.log
run() {  # name, extra args
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c7_bench_$1.json 2> gpurun_out/r3c7_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c7_bench_$1.json"))
r=d["roofline"]
print("%-22s ms/step %.2f loss %.5f gemm union %.2f" % ("$1", d["ms_per_step"], d["loss"], r["gemm_ms_per_step"]))
PY
}
run sw0_a "--opt 2=0"
run sw2_a "--opt 2=2"
run sw1_a "--opt 2=1"
run sw0_b "--opt 2=0"
run sw2_b "--opt 2=2"
run sw2_onechain "--opt 2=2,11=0"
run sw0_onechain "--opt 2=0,11=0"
