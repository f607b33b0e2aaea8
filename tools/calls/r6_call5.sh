#!/bin/bash
# round 6, call 5: LoRA up-projections in the q|k|v GEMM epilogues (option 22) + greedy bookkeeping in one launch (uvx_greedy_select): tests, step flavours old vs new
# on one box, decode lines (8B and 70B, B = 1 / 8) with and without the one-launch bookkeeping
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c5; mkdir -p $O
timeout 1200 python -m pytest tests/test_lora_gpu.py tests/test_generate_gpu.py tests/test_kl_gpu.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gelu or mfma or attention or gemm" 2>&1 | tail -3 | tee -a $O/pytest.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4))"; }
for rep in 1 2; do
for f in "lora8_old:--audio-lora-r 8 --opt 19=1,20=1,21=1,22=1" "lora8_lora_separate:--audio-lora-r 8 --opt 22=1" "lora8_new:--audio-lora-r 8" "kl_lora8_old:--loss kl --audio-lora-r 8 --opt 19=1,20=1,21=1,22=1" "kl_lora8_new:--loss kl --audio-lora-r 8" "ce_new:"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>/dev/null | tail -1 | line $name | tee -a $O/flavours.txt
done
done
dline() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', 'decode ms/token', round(r['decode_ms_per_token'],3), 'frac_hbm', round(r['roofline']['frac'],4), 'prefill ms', round(r['prefill_ms'],2), 'tok/s', round(r['value'],1))"; }
for rep in 1 2; do
for f in "c4s_b1_generic:UVX_GREEDY_SELECT=0:--workload c4s --batch 1" "c4s_b1_select:UVX_GREEDY_SELECT=1:--workload c4s --batch 1" "c4s_b8_generic:UVX_GREEDY_SELECT=0:--workload c4s --batch 8" "c4s_b8_select:UVX_GREEDY_SELECT=1:--workload c4s --batch 8"; do
  name=${f%%:*}; rest=${f#*:}; envv=${rest%%:*}; flags=${rest#*:}
  env $envv timeout 600 python bench.py $flags --steps 3 --warmup 1 2>/dev/null | tail -1 | dline $name | tee -a $O/decode.txt
done
done
for f in "c4_b1_generic:UVX_GREEDY_SELECT=0:--workload c4 --batch 1" "c4_b1_select:UVX_GREEDY_SELECT=1:--workload c4 --batch 1"; do
  name=${f%%:*}; rest=${f#*:}; envv=${rest%%:*}; flags=${rest#*:}
  env $envv timeout 900 python bench.py $flags --steps 3 --warmup 1 2> $O/$name.err | tail -1 > $O/bench_$name.json; cat $O/bench_$name.json | dline $name | tee -a $O/decode.txt
done
