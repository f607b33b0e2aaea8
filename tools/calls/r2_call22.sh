#!/bin/bash
# Round 2, call 22: in-situ A/B of the tile choice for the other single-round dgrad shapes (160-row picked; 192-row alternative).
R=$PWD; OUT=$R/gpurun_out/r2c22; mkdir -p $OUT; export PYTHONPATH=$R
for i in 1 2; do
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/base_$i.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --gemm-override 2528x4096x6144=32 > $OUT/k6144_v32_$i.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --gemm-override 2528x4096x4096=32 > $OUT/k4096_v32_$i.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --gemm-override 2528x4096x14336=32 > $OUT/k14336_v32_$i.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --gemm-override 12000x1024x1024=31 > $OUT/enc1024_v31_$i.log 2>&1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.log")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["ms_per_step"], 2), round(j["roofline"]["achieved"], 1))
    except Exception as e: print(f, "failed", e)
PY
