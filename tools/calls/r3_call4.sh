#!/bin/bash
# round 3, call 4: staggered two-stream schedule (option 11 = 3) and tile picks for the half-batch GEMMs (in-situ A/B), epilogue cost
# probe for the encoder GEMMs, the new log-mel fixture tests
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -k "logmel or two_stream" > gpurun_out/r3c4_tests.log 2>&1
tail -6 gpurun_out/r3c4_tests.log
python tools/gpu_gemm_epilogue_probe.py > gpurun_out/r3c4_epilogue_probe.txt 2>&1
cat gpurun_out/r3c4_epilogue_probe.txt
V31="1264x4096x28672=31,1264x4096x14336=31,1264x14336x4096=31,1264x4096x4096=31,1264x4096x6144=31,1264x6144x4096=31"
VWIDE="1264x14336x4096=31,1264x6144x4096=31"
V32="1264x4096x28672=32,1264x4096x14336=32,1264x4096x4096=32,1264x4096x6144=32,1264x14336x4096=31,1264x6144x4096=31"
run() {  # name, extra args
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c4_bench_$1.json 2> gpurun_out/r3c4_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c4_bench_$1.json"))
r=d["roofline"]
print("%-22s ms/step %.2f loss %.5f gemm union %.2f TF/s %.1f" % ("$1", d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["achieved"]))
PY
}
run lockstep_a "--opt 11=2"
run stagger_a "--opt 11=3"
run stagger_v31 "--opt 11=3 --gemm-override $V31"
run stagger_wide31 "--opt 11=3 --gemm-override $VWIDE"
run stagger_v32 "--opt 11=3 --gemm-override $V32"
run lockstep_b "--opt 11=2"
run stagger_b "--opt 11=3"
run lockstep_wide31 "--opt 11=2 --gemm-override $VWIDE"
