#!/bin/bash
# round 6, call 1: head_dim-64 attention tile forms (options 19 / 20): tests, the kernel probe, step flavours old vs new on one box,
# and the kernel statistics of the LoRA-only flavour (is the training tower's 171-us forward attention a concurrency artefact of the KL side stream?)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c1; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -5 | tee $O/pytest_attention.txt
timeout 300 python tools/gpu_attn_r6_probe.py 2 2>&1 | tee $O/attn_probe.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4))"; }
for rep in 1 2; do
for f in "ce_old:--opt 19=1,20=1" "ce_new:" "kl_lora8_old:--loss kl --audio-lora-r 8 --opt 19=1,20=1" "kl_lora8_new:--loss kl --audio-lora-r 8"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>/dev/null | tail -1 | line $name | tee -a $O/flavours.txt
done
done
cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-live-traffic --audio-lora-r 8 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/stats/s_kernel_stats.csv")))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
mine = [r for r in rows if "at::native" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in mine) / 4 / 1e6
with open("$O/kernel_stats_lora8.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --audio-lora-r 8 (4 steps incl. warm-up): {tot:.1f} ms of kernels per step\n")
    for r in sorted(mine, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
        ms = float(r["TotalDurationNs"]) / 4 / 1e6
        f.write(f"{short(r['Name'])[:70]:70s} {int(r['Calls']) / 4:8.1f} {ms:8.3f} {float(r['AverageNs']) / 1e3:9.1f} {100 * ms / tot:6.2f}\n")
PY
head -30 $O/kernel_stats_lora8.txt
rm -rf $O/stats
