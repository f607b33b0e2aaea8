#!/bin/bash
# Round 2, call 23: in-situ A/B of the tail split (second launch for the last partial round) and of the persistent probe builds
# on the encoder's short-K shapes.
R=$PWD; OUT=$R/gpurun_out/r2c23; mkdir -p $OUT; export PYTHONPATH=$R
for i in 1 2; do
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/base_$i.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --opt 10=100 > $OUT/split100_$i.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --opt 10=1 > $OUT/nosplit_$i.log 2>&1
  UVX_LIB=$R/ultravox_amd/libuvx_probes.so timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --gemm-override 12000x4096x1024=35,12000x3072x1024=36,12000x1024x1024=36 > $OUT/persist_enc_$i.log 2>&1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.log")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["ms_per_step"], 2), round(j["roofline"]["achieved"], 1), j["roofline"]["launches_per_step"])
    except Exception as e: print(f, "failed", e)
PY
