#!/bin/bash
# round 3, call 2: whole GPU suite on the restructured LLM orchestration (two-stream schedule, llm_act) + same-box A/B of option 11
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --deselect tests/test_c2_full_depth_gpu.py > gpurun_out/r3c2_tests.log 2>&1
tail -15 gpurun_out/r3c2_tests.log
for rep in 1 2; do
  for o in 0 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt 11=$o --gemm-table gpurun_out/r3c2_gemm_table_opt11_${o}_rep${rep}.txt > gpurun_out/r3c2_bench_opt11_${o}_rep${rep}.json 2> gpurun_out/r3c2_bench_opt11_${o}_rep${rep}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r3c2_bench_opt11_${o}_rep${rep}.json"))
r=d["roofline"]
print("opt11=$o rep$rep ms/step %.2f loss %.5f gemm union %.2f summed %.2f TF/s %.1f" % (d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["gemm_ms_per_step_summed_intervals"], r["achieved"]))
PY
  done
done
