#!/bin/bash
# round 6, call 9: where the 8B model's decode token goes (rocprofv3 kernel trace of generate(): 1 prompt, 33 tokens), and the SwiGLU-backward epilogue (option 2) on the clean build
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6c9; mkdir -p $O
cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 33 > $O/decode_probe.log 2>&1
tail -2 $O/decode_probe.log
cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/stats/s_kernel_stats.csv")))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
with open("$O/decode8_b1_kernel_stats.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- tools/gpu_decode_probe.py 1 33 (Llama-3-8B, B = 1: 2 warm-up + 2 timed generate() calls: 4 prefills, 2 x 32 + 2 x 0 decode steps... totals over the process): {tot:.1f} ms of kernels\n")
    for r in rows[:40]:
        f.write(f"{short(r['Name'])[:80]:80s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:9.3f} {float(r['AverageNs']) / 1e3:9.1f} {100 * float(r['TotalDurationNs']) / 1e6 / tot:6.2f}\n")
print(open("$O/decode8_b1_kernel_stats.txt").read())
PY
rm -rf $O/stats
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'loss', round(r['loss'],4), 'gemm_ms', round(r['roofline']['gemm_ms_per_step'],2))"; }
for rep in 1 2; do
for f in "ce:" "ce_swiglu_bwd_epilogue:--opt 2=2"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>/dev/null | tail -1 | line $name | tee -a gpurun_out/r6c9/swiglu_bwd_ab.txt
done
done
