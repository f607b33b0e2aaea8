#!/bin/bash
# round 4, call 29: KL distillation with the teacher pass on a side stream next to the student forward (same kernels): same-box A/B, KL and KL + encoder LoRA
# (as issued: the switch was --kl-side-stream, off by default; since call 30 the side stream is the default and --no-kl-side-stream turns it off)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c29; mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
run() { timeout 200 $B $2 > $O/bench_$1.json 2>/dev/null; python - <<PY
import json
r=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("$1".ljust(18), "ms/step", round(r["ms_per_step"],2), "value", round(r["value"],1), "loss", r["loss"])
PY
}
run ce ""
run kl_a "--loss kl"
run kl_side_a "--loss kl --kl-side-stream"
run kl_b "--loss kl"
run kl_side_b "--loss kl --kl-side-stream"
run kl_lora "--loss kl --audio-lora-r 8"
run kl_lora_side "--loss kl --audio-lora-r 8 --kl-side-stream"
