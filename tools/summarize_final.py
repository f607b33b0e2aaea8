"""(CPU) Turn the raw outputs of tools/final_profile.sh (gpurun_out/final/) into the committed round summaries:
   profiles/rNN_kernel_stats_final.txt  - rocprofv3 --kernel-trace --stats, per step, libuvx kernels
   profiles/rNN_pmc_traffic.json        - memory-side bytes per GEMM launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE)
   profiles/rNN_bench_c2.json, rNN_gemm_table_insitu_final.txt
usage: python tools/summarize_final.py r02 [steps_profiled=4]"""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4          # bench.py --steps 3 --warmup 1 under the profiler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]

rows = list(csv.DictReader(open(os.path.join(src, "stats", "s_kernel_stats.csv"))))
mine = [r for r in rows if "at::native" not in r["Name"] and "rccl" not in r["Name"].lower()]
total = sum(float(r["TotalDurationNs"]) for r in mine) / steps / 1e6
bench = json.loads([l for l in open(os.path.join(src, "bench.log")) if l.startswith("{")][-1])
with open(os.path.join(dst, f"{tag}_kernel_stats_final.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof   (C2; {steps} steps incl. warm-up)\n")
    f.write(f"# libuvx kernels only: {total:.1f} ms per step under the profiler (bench.py on the same box: {bench['ms_per_step']:.1f} ms/step)\n")
    f.write(f"{'kernel':64s} {'calls/step':>10s} {'ms/step':>9s} {'avg_us':>9s} {'pct':>6s}\n")
    for r in sorted(mine, key=lambda r: -float(r["TotalDurationNs"])):
        ms = float(r["TotalDurationNs"]) / steps / 1e6
        f.write(f"{short(r['Name'])[:64]:64s} {int(r['Calls']) / steps:10.1f} {ms:9.3f} {float(r['AverageNs']) / 1e3:9.1f} {100 * ms / total:6.2f}\n")
    gemm = sum(float(r["TotalDurationNs"]) for r in mine if "gemm_nt" in r["Name"] or "gemm_skinny" in r["Name"]) / steps / 1e6
    f.write(f"# GEMM kernels {gemm:.2f} ms/step; everything else {total - gemm:.2f} ms/step\n")

tr = json.load(open(os.path.join(src, "pmc_traffic.json")))
g = lambda d: [(v[0], v[1]) for k, v in d.items() if k.startswith("gemm_nt")]
nf, sf = map(sum, zip(*g(tr["fetch"])))
nw, sw = map(sum, zip(*g(tr["write"])))
fetch_kb, write_kb = sf / nf, sw / nw
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof",
       "kernel": "gemm_nt_bf16_* (all tile variants)", "launches": int(nf), "fetch_kb_per_launch_raw": fetch_kb,
       "write_kb_per_launch": write_kb, "gfx950_fetch_correction": 2.0,
       "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0, "algorithmic_bytes_per_launch": alg,
       "note": "FETCH_SIZE / WRITE_SIZE are memory-side (fabric) counters: Infinity-Cache hits are included, so this is L2-miss traffic, "
               "an upper bound on HBM bytes; FETCH_SIZE x 2 per MI355X_MICROARCH.md (gfx950 counts 128-byte requests at 64 bytes)"}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
json.dump(bench, open(os.path.join(dst, f"{tag}_bench_c2.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "gemm_table.txt"), os.path.join(dst, f"{tag}_gemm_table_insitu_final.txt"))
print(f"{tag}: {total:.1f} ms/step of kernels ({gemm:.1f} GEMM), bench {bench['ms_per_step']:.1f} ms/step, GEMM traffic "
      f"{out['traffic_bytes_per_launch'] / 1e6:.0f} MB/launch vs {alg / 1e6:.0f} MB algorithmic")
