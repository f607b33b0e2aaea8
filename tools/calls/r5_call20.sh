#!/bin/bash
# round 5, call 20: the reference's real recipe flavour (KL distillation + rank-8 encoder LoRA) on the final build: bench lines (CE, LoRA, KL, KL + LoRA, one box)
# and the rocprofv3 kernel statistics of the KL + LoRA step
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c20; mkdir -p $O
for f in "ce:" "lora8:--audio-lora-r 8" "kl:--loss kl" "kl_lora8:--loss kl --audio-lora-r 8"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline $flags 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$name ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'mfu', round(r['mfu'],4), 'loss', round(r['loss'],4))" | tee -a $O/flavours.txt
done
cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --loss kl --audio-lora-r 8 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/stats/s_kernel_stats.csv")))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
mine = [r for r in rows if "at::native" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in mine) / 4 / 1e6
with open("$O/kernel_stats_kl_lora8.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --loss kl --audio-lora-r 8 (4 steps incl. warm-up): {tot:.1f} ms of kernels per step\n")
    for r in sorted(mine, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
        ms = float(r["TotalDurationNs"]) / 4 / 1e6
        f.write(f"{short(r['Name'])[:70]:70s} {int(r['Calls']) / 4:8.1f} {ms:8.3f} {float(r['AverageNs']) / 1e3:9.1f} {100 * ms / tot:6.2f}\n")
print(open("$O/kernel_stats_kl_lora8.txt").read())
PY
rm -rf $O/stats
