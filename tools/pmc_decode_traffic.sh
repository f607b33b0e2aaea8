#!/bin/bash
# Memory-side read traffic of the 70B decode GEMVs (rocprofv3 --pmc FETCH_SIZE, its own pass, --kernel-trace only) against the weight bytes they
# must stream -> gpurun_out/pmc_decode/summary.json (copied to profiles/rNN_pmc_decode_traffic.json; bench.py --workload c4 reports it as roofline.traffic)
R=$PWD; OUT=$R/gpurun_out/pmc_decode; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- timeout 500 python $R/tools/gpu_decode_probe.py 1 6 meta-llama/Llama-3.3-70B-Instruct > $OUT/log.txt 2>&1
python - <<PY
import csv, collections, json, os
f = "$OUT/fetch/p_counter_collection.csv"
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    gx = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0); wx = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)
    a = agg[(k, gx // max(wx, 1))]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {}
for (k, blocks), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    out[f"{k} [{blocks} blocks]"] = {"launches": n, "fetch_kb_per_launch_raw": s / n, "fetch_mb_per_launch_x2": 2 * s / n / 1024}
    print(f"{k:60s} blocks {blocks:6d} launches {n:6d} FETCH_SIZE x2 per launch {2 * s / n / 1024:10.1f} MB")
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
os.remove(f)
for fn in os.listdir("$OUT/fetch"):
    if fn.endswith("kernel_trace.csv"): os.remove(os.path.join("$OUT/fetch", fn))
PY
