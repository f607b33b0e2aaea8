#!/bin/bash
# round 3, call 39: fused q+k LoRA pass (NOT ADOPTED - kernels and option 16 were removed again; profiles/r03_lora_fused_qk_negative.txt): tests, then the encoder-LoRA flavour with the fused and separate kernels, same box
timeout 900 python -m pytest tests/test_lora_gpu.py tests/test_qwen_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -4
run() { timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --audio-lora-r 8 $2 > gpurun_out/r3c39_$1.json 2> gpurun_out/r3c39_$1.err
  python -c "
import json; d=json.load(open('gpurun_out/r3c39_$1.json')); print('%-12s ms/step %.2f loss %.5f' % ('$1', d['ms_per_step'], d['loss']))" || tail -3 gpurun_out/r3c39_$1.err; }
run warm ""
run fused ""
run separate "--opt 16=1"
run fused_b ""
run separate_b "--opt 16=1"
