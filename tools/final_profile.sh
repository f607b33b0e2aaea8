#!/bin/bash
# Round-end evidence run on the GPU box: tests, bench JSON, rocprofv3 kernel stats, PMC traffic passes.
R=$PWD; OUT=$R/gpurun_out/final; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 --gemm-table $OUT/gemm_table.txt > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --autotune > $OUT/bench_autotune.log 2>&1; tail -1 $OUT/bench_autotune.log | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --opt 11=2,13=0 > $OUT/bench_two_chains.log 2>&1; tail -1 $OUT/bench_two_chains.log | cut -c1-200
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $OUT/rocprof_stats.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $OUT/rocprof_fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $OUT/rocprof_write.log 2>&1
ls $OUT/stats $OUT/fetch $OUT/write | head -30
# keep only what is needed (the merged-back directory is capped at 64 MiB)
python - <<PY
import csv, collections, json, os
out = "$OUT"
res = {}
for name in ("fetch", "write"):
    f = os.path.join(out, name, "p_counter_collection.csv")
    if not os.path.exists(f): continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"])
    res[name] = {k: v for k, v in agg.items()}
    os.remove(f)
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=0)
for name in ("fetch", "write", "stats"):
    d = os.path.join(out, name)
    for fn in os.listdir(d):
        if fn.endswith("kernel_trace.csv") and name != "stats": os.remove(os.path.join(d, fn))
PY
du -sh $OUT
