#!/bin/bash
# round 3, call 24: staged (whole-row-segment) stores in every bf16 attention epilogue: tests, then the per-kernel times of a step
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -3
PYTHONPATH=. timeout 300 python tools/gpu_attn_timeline.py 2>&1 | grep -v amdgpu.ids | head -3
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c24_bench_$i.json 2> gpurun_out/r3c24_bench_$i.err
python -c "
import json; d=json.load(open('gpurun_out/r3c24_bench_$i.json')); print('bench $i ms/step %.2f loss %.5f' % (d['ms_per_step'], d['loss']))"
done
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3c24_stats -o s --output-format csv -- timeout 300 python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3c24_stats/**/s_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    if "gemm" in n: continue
    print("%-72s calls %6s avg_us %9.1f total_ms %9.2f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf gpurun_out/r3c24_stats
