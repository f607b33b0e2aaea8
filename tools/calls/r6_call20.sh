#!/bin/bash
# round 6, call 20: the KL rows path with the last layer on the loss rows only (student + teacher): tests, flavour A/B (option 3 = 0: full-row last layer AND full-logit CE heads - the KL path's heads stay on rows)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c20; mkdir -p $O
timeout 1200 python -m pytest tests/test_kl_gpu.py tests/test_model_gpu.py tests/test_lora_gpu.py tests/test_checkpoint_gpu.py -q 2>&1 | tail -8 | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4), 'mfu', round(r['mfu'],4))"; }
for rep in 1 2; do
for f in "kl_full_last_layer:--loss kl --opt 3=0" "kl_compact:--loss kl" "kl_lora8_full_last_layer:--loss kl --audio-lora-r 8 --opt 3=0" "kl_lora8_compact:--loss kl --audio-lora-r 8" "ce:"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>$O/$name.err | tail -1 | line $name | tee -a $O/flavours.txt
done
done
