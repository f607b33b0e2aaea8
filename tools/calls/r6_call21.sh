#!/bin/bash
# round 6, call 21: LoRA target_modules beyond q / k (ABI 17: uvx_enc_lora_layer_t.v / .o) - the new tests, then every test that touches the adapters, the KL
# step, the model and the checkpoints (the default q / k path goes through the restructured forward / backward), then the recipe flavours for the record
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c21; mkdir -p $O
timeout 900 python -m pytest tests/test_lora_gpu.py -q -x -k "target_modules or v_and_o or gemma3" 2>&1 | tail -15 | tee $O/pytest_new.txt
timeout 1500 python -m pytest tests/test_lora_gpu.py tests/test_kl_gpu.py tests/test_model_gpu.py tests/test_checkpoint_gpu.py tests/test_gemma3_gpu.py tests/test_qwen_gpu.py tests/test_baseline_configs_gpu.py -q 2>&1 | tail -8 | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4), 'mfu', round(r['mfu'],4))"; }
for f in "ce:" "lora8:--audio-lora-r 8" "kl_lora8:--loss kl --audio-lora-r 8"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>$O/$name.err | tail -1 | line $name | tee -a $O/flavours.txt
done
