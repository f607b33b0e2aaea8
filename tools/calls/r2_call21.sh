#!/bin/bash
# Round 2, call 21: in-situ A/B of the tile choice for 2528 x 4096 x 28672 (the gate|up dgrad): 192-row (picked) vs 160-row tile.
R=$PWD; OUT=$R/gpurun_out/r2c21; mkdir -p $OUT; export PYTHONPATH=$R
for i in 1 2; do
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/base_$i.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --gemm-override 2528x4096x28672=33 > $OUT/v33_$i.log 2>&1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.log")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["ms_per_step"], 2), round(j["roofline"]["achieved"], 1))
    except Exception as e: print(f, "failed", e)
PY
