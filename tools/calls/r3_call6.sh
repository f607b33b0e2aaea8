#!/bin/bash
# round 3, call 6: where the step's time goes inside the two-chain schedule (skip-ablation probe, option 15), SwiGLU-backward epilogue
# fusion re-tested under two chains (option 2), epilogue / cold-operand probe of the GEMM shapes
mkdir -p gpurun_out
run() {  # name, extra args
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c6_bench_$1.json 2> gpurun_out/r3c6_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c6_bench_$1.json"))
r=d["roofline"]
print("%-22s ms/step %.2f gemm union %.2f" % ("$1", d["ms_per_step"], r["gemm_ms_per_step"]))
PY
}
run base_a ""
run swiglu_fused "--opt 2=1"
run skip_attn_bwd "--probe-skip 1"
run skip_attn_fwd "--probe-skip 2"
run skip_swiglu_bwd "--probe-skip 4"
run skip_rms_bwd "--probe-skip 8"
run skip_rms_fwd "--probe-skip 16"
run skip_rope_fwd "--probe-skip 32"
run skip_enc_attn "--probe-skip 64"
run skip_enc_ln "--probe-skip 128"
run skip_all "--probe-skip 255"
run base_b ""
run swiglu_fused_b "--opt 2=1"
python tools/gpu_gemm_epilogue_probe.py > gpurun_out/r3c6_epilogue_probe.txt 2>&1
cat gpurun_out/r3c6_epilogue_probe.txt
