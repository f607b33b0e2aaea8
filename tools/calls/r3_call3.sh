#!/bin/bash
# round 3, call 3: transposing-LDS-read attention kernels (option 12): probe of the instruction's semantics, attention tests, whole
# model tests, then same-box A/B of option 12 inside bench.py (two-stream schedule on in both arms)
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "lds_transpose or attention" > gpurun_out/r3c3_tests_attn.log 2>&1
tail -15 gpurun_out/r3c3_tests_attn.log
python -m pytest tests -m gpu -q --deselect tests/test_c2_full_depth_gpu.py > gpurun_out/r3c3_tests.log 2>&1
tail -8 gpurun_out/r3c3_tests.log
for rep in 1 2; do
  for o in 0 1; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt 12=$o > gpurun_out/r3c3_bench_opt12_${o}_rep${rep}.json 2> gpurun_out/r3c3_bench_opt12_${o}_rep${rep}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r3c3_bench_opt12_${o}_rep${rep}.json"))
r=d["roofline"]
print("opt12=$o rep$rep ms/step %.2f loss %.5f gemm union %.2f TF/s %.1f" % (d["ms_per_step"], d["loss"], r["gemm_ms_per_step"], r["achieved"]))
PY
  done
done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3c3_prof -o r3c3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $GRAFT_REPO_ROOT/gpurun_out/r3c3_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3c3_prof.err
ls $GRAFT_REPO_ROOT/gpurun_out/r3c3_prof | head
