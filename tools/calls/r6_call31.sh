#!/bin/bash
# round 6, call 31: the backward from the first audio token on the Qwen3 / Qwen2 / Gemma-3 backbones (q / k norm backward through the row map; the entry without a
# compact last layer) - tests of everything the change touches, then the q3 / g3 recipe lines with and without it
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c31; mkdir -p $O
timeout 900 python -m pytest tests/test_prefix_skip_gpu.py tests/test_qwen_gpu.py tests/test_gemma3_gpu.py tests/test_gemma_gpu.py tests/test_model_gpu.py tests/test_kl_gpu.py -q -x 2>&1 | tail -8 | tee $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4), 'mfu', round(r['mfu'],4), 'from', r['config']['llm_backward_from_position'])"; }
for wl in q3 g3; do
  for arm in "full:--no-prefix-skip" "from_first_audio:"; do
    an=${arm%%:*}; af=${arm#*:}
    timeout 900 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic $af 2>$O/$wl.$an.err | tail -1 | line "$wl $an" | tee -a $O/backbones_prefix_skip_ab.txt
  done
done
