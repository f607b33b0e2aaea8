#!/bin/bash
# round 4, call 19: which memory-side counters exist (can HBM reads be told apart from Infinity-Cache hits?) + a pass with the DRAM-qualified request counters
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4c19; mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -i "dram\|mall\|hbm\|RDREQ\|WRREQ\|EA0\|EA_" $O/counters.txt | cut -c1-200 | head -60
for c in TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum; do
  if grep -q "$c" $O/counters.txt; then
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- timeout 200 python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/log_$c.txt 2>&1
    python - <<PY
import csv, collections
f="$O/$c/p_counter_collection.csv"
try:
    n=0; s=0.0
    for r in csv.DictReader(open(f)):
        if "gemm_nt" in r["Kernel_Name"]: n+=1; s+=float(r["Counter_Value"])
    print("$c", "gemm launches", n, "per launch", s/max(n,1))
except Exception as e: print("$c", "failed", e)
PY
    rm -rf $O/$c
  else echo "$c: not listed"; fi
done
