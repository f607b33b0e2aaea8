#!/bin/bash
# HBM traffic of every kernel of the C2 step from the memory-side PMC counters (separate passes: FETCH_SIZE uses 3
# of the 4 TCC slots, WRITE_SIZE 2).  Output: gpurun_out/final/pmc_traffic.json {pass: {kernel: [launches, sum]}}.
R=$PWD; OUT=$R/gpurun_out/final; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $OUT/rocprof_fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $OUT/rocprof_write.log 2>&1
python - <<PY
import csv, collections, json, os
out = "$OUT"
res = {}
for name in ("fetch", "write"):
    f = os.path.join(out, name, "p_counter_collection.csv")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"])
    res[name] = dict(agg)
    os.remove(f); os.remove(os.path.join(out, name, "p_kernel_trace.csv"))
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=0)
for name in res:
    g = {k: v for k, v in res[name].items() if k.startswith("gemm_nt")}
    n = sum(v[0] for v in g.values()); s = sum(v[1] for v in g.values())
    print(name, "gemm launches", n, "counter sum", s, "per launch", s / max(n, 1))
PY
