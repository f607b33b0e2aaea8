#!/bin/bash
# round 3, call 36: kernel stats of the encoder-LoRA step flavour (the reference's release recipes train rank-8 LoRA on the Whisper q / k projections)
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3c36_stats -o s --output-format csv -- timeout 300 python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --audio-lora-r 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3c36_stats/**/s_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    if n.startswith("at::") or "rocclr" in n: continue
    ms = float(r["TotalDurationNs"]) / 1e6 / 4
    tot += ms
    if ms > 0.05: print("%-72s calls/step %7.1f avg_us %9.1f ms/step %8.3f" % (n, int(r["Calls"]) / 4, float(r["AverageNs"]) / 1e3, ms))
print("total ms/step of libuvx kernels:", round(tot, 2))
PY
rm -rf gpurun_out/r3c36_stats
