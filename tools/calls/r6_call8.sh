#!/bin/bash
# round 6, call 8: the workload lines again on the spill-free build (call 4's c3 / c5 / q3 / g3 / l70 figures were taken while the 256-row tile carried a scratch segment),
# the inference lines (c4 B = 1 / 8, c4s B = 1), and the KL + LoRA flavour's kernel statistics
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c8; mkdir -p $O
for w in c3 c5 q3 g3 l70; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline 2> $O/bench_$w.err | tail -1 > $O/bench_$w.json
  python -c "
import json; r = json.loads(open('$O/bench_$w.json').read()); print('$w', {k: r.get(k) for k in ('value','ms_per_step','mfu')}, (r.get('roofline') or {}).get('frac'))" || tail -5 $O/bench_$w.err
done
for f in "c4_b1:--workload c4 --batch 1" "c4_b8:--workload c4 --batch 8" "c4s_b1:--workload c4s --batch 1"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 900 python bench.py $flags 2> $O/$name.err | tail -1 > $O/bench_$name.json
  python -c "
import json; r = json.loads(open('$O/bench_$name.json').read()); print('$name', 'decode ms/token', round(r['decode_ms_per_token'],3), 'frac_hbm', round(r['roofline']['frac'],4), 'prefill ms', round(r['prefill_ms'],2), 'prefill frac_mfma', round(r['prefill']['frac_mfma'],3))"
done
cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-live-traffic --loss kl --audio-lora-r 8 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
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
PY
head -12 $O/kernel_stats_kl_lora8.txt; rm -rf $O/stats
