#!/bin/bash
# round 3, call 44: the reference's actual v0.6 recipe flavour (meta_config.yaml: KL-divergence loss; audio_model_lora_config r = 8) on the three
# v0.6 backbones, full depth, 8 x 30 s clips, ONE MI355X each
run() { timeout 900 python bench.py --workload $1 --steps 4 --warmup 2 --no-cpu-baseline --loss kl --audio-lora-r 8 > gpurun_out/r3c44_$1.json 2> gpurun_out/r3c44_$1.err
  python -c "
import json; d=json.load(open('gpurun_out/r3c44_$1.json')); print('%-5s ms/step %.1f value %.1f audio-s/s mfu %.4f gemm frac %.4f loss %.5f' % ('$1', d['ms_per_step'], d['value'], d['mfu'], d['roofline']['frac'], d['loss']))" || tail -5 gpurun_out/r3c44_$1.err; }
run q3; run g3; run l70
