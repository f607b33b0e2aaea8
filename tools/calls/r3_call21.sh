#!/bin/bash
# round 3, call 21: the other step flavours and workloads on the final build, one box
mkdir -p gpurun_out/r03_flavours
run() {  # name, args
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r03_flavours/$1.json 2> gpurun_out/r03_flavours/$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03_flavours/$1.json"))
print("%-22s ms/step %.2f value %.1f mfu %.4f loss %.4f" % ("$1", d["ms_per_step"], d["value"], d["mfu"], d["loss"]))
PY
}
run c2_ce ""
run c2_kl "--loss kl"
run c2_audio_lora8 "--audio-lora-r 8"
run c2_kl_audio_lora8 "--loss kl --audio-lora-r 8"
run c3 "--workload c3"
run c5 "--workload c5"
run c2_ce_b ""
