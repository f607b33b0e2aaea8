#!/bin/bash
# round 6, call 32: AttnBwdDesc::d_first in the dQ + dK/dV kernel pair (T > 320, head_dim 64 / 256, sliding windows): tests of everything that runs an attention
# backward, then the C5 line (Gemma-7B, head_dim 256 + wav2vec2-large) and an encoder-LoRA flavour (the pair is the Whisper tower's backward) with and without the prefix skip
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c32; mkdir -p $O
timeout 1500 python -m pytest tests/test_prefix_skip_gpu.py tests/test_kernels_gpu.py tests/test_lora_gpu.py tests/test_gemma_gpu.py tests/test_gemma3_gpu.py tests/test_qwen_gpu.py tests/test_model_gpu.py tests/test_kl_gpu.py tests/test_wav2vec2_gpu.py tests/test_baseline_configs_gpu.py -q 2>&1 | tail -8 | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4), 'mfu', round(r['mfu'],4), 'from', r['config']['llm_backward_from_position'])"; }
for f in "c5:--workload c5" "c2_lora8:--audio-lora-r 8"; do
  name=${f%%:*}; flags=${f#*:}
  for arm in "full:--no-prefix-skip" "from_first_audio:"; do
    an=${arm%%:*}; af=${arm#*:}
    timeout 900 python bench.py $flags --steps 8 --warmup 3 --no-cpu-baseline --no-live-traffic $af 2>$O/$name.$an.err | tail -1 | line "$name $an" | tee -a $O/pair_prefix_skip_ab.txt
  done
done
