#!/bin/bash
# round 6, call 17: the q_proj / k_proj adapter products as paired launches (lora_down2 / lora_up2 / one partial-sum launch for the four weight gradients; option 22 = 1: separate)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c17; mkdir -p $O
timeout 900 python -m pytest tests/test_lora_gpu.py tests/test_kl_gpu.py tests/test_checkpoint_gpu.py -q 2>&1 | grep -E "passed|failed|FAILED|rror" | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4))"; }
for rep in 1 2; do
for f in "lora8_separate:--audio-lora-r 8 --opt 22=1" "lora8_paired:--audio-lora-r 8" "kl_lora8_separate:--loss kl --audio-lora-r 8 --opt 22=1" "kl_lora8_paired:--loss kl --audio-lora-r 8"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>/dev/null | tail -1 | line $name | tee -a $O/flavours.txt
done
done
