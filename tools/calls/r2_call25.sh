#!/bin/bash
# Round 2, call 25: bench lines of the other step flavours on the final build (encoder LoRA r = 8, KL-distillation loss).
R=$PWD; OUT=$R/gpurun_out/r2c25; mkdir -p $OUT; export PYTHONPATH=$R
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/base.log 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --audio-lora-r 8 > $OUT/lora_r8.log 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --loss kl > $OUT/kl.log 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.log")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["ms_per_step"], 2), round(j["value"], 1), round(j["roofline"]["achieved"], 1))
    except Exception as e: print(f, "failed", e, open(f).read()[-300:])
PY
