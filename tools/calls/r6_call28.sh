#!/bin/bash
# round 6, call 28: uvx_llm_bwd_rows_from (the KL recipe's backward from the first audio token) - tests, then the recipe flavours with and without the prefix skip
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c28; mkdir -p $O
timeout 600 python -m pytest tests/test_prefix_skip_gpu.py tests/test_kl_gpu.py tests/test_lora_gpu.py -q -x 2>&1 | tail -8 | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4), 'mfu', round(r['mfu'],4), 'from', r['config']['llm_backward_from_position'])"; }
for f in "lora8:--audio-lora-r 8" "kl:--loss kl" "kl_lora8:--loss kl --audio-lora-r 8"; do
  name=${f%%:*}; flags=${f#*:}
  for arm in "full:--no-prefix-skip" "from_first_audio:"; do
    an=${arm%%:*}; af=${arm#*:}
    timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags $af 2>$O/$name.$an.err | tail -1 | line "$name $an" | tee -a $O/flavours_prefix_skip_ab.txt
  done
done
