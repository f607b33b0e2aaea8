#!/bin/bash
# round 3, call 9: RMSNorm fused across the GEMMs (option 13) - whole GPU suite, then in-situ A/B (x one / two chains), the trainer's
# schedule auto-tuner
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --deselect tests/test_c2_full_depth_gpu.py > gpurun_out/r3c9_tests.log 2>&1
tail -12 gpurun_out/r3c9_tests.log
run() {  # name, extra args
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c9_bench_$1.json 2> gpurun_out/r3c9_bench_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c9_bench_$1.json"))
r=d["roofline"]
print("%-26s ms/step %.2f loss %.5f gemm union %.2f TF/s %.1f streams %s %s" % ("$1", d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["achieved"], d["llm_streams"], d.get("llm_streams_autotuned_ms")))
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
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3c9_bench_autotune.json 2> gpurun_out/r3c9_bench_autotune.err
python - <<PY
import json
d=json.load(open("gpurun_out/r3c9_bench_autotune.json"))
print("autotuned: ms/step %.2f streams %s timings %s mfu %.4f" % (d["ms_per_step"], d["llm_streams"], d.get("llm_streams_autotuned_ms"), d["mfu"]))
PY
