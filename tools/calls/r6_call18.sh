#!/bin/bash
# round 6, call 18: RMSNorm inside the staged weight-streaming kernel at 3..16 decode rows (option 24 = 1: the two launches): tests, decode lines A/B
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c18; mkdir -p $O
timeout 900 python -m pytest tests/test_generate_gpu.py tests/test_kernels_gpu.py -q -k "rmsnorm or decode or generate" 2>&1 | tail -8 | tee $O/pytest.txt
dline() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', 'decode ms/token', round(r['decode_ms_per_token'],3), 'frac_hbm', round(r['roofline']['frac'],4), 'prefill ms', round(r['prefill_ms'],2), 'tok/s', round(r['value'],1))"; }
for rep in 1 2; do
for f in "c4s_b4_two:--workload c4s --batch 4 --opt 24=1" "c4s_b4_fused:--workload c4s --batch 4" "c4s_b8_two:--workload c4s --batch 8 --opt 24=1" "c4s_b8_fused:--workload c4s --batch 8" "c4s_b16_two:--workload c4s --batch 16 --opt 24=1" "c4s_b16_fused:--workload c4s --batch 16"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 600 python bench.py $flags --steps 3 --warmup 1 2>$O/$name.err | tail -1 | dline $name | tee -a $O/decode.txt
done
done
for f in "c4_b8_two:--workload c4 --batch 8 --opt 24=1" "c4_b8_fused:--workload c4 --batch 8" "c4_b16_two:--workload c4 --batch 16 --opt 24=1" "c4_b16_fused:--workload c4 --batch 16" "c4_b4_two:--workload c4 --batch 4 --opt 24=1" "c4_b4_fused:--workload c4 --batch 4"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 900 python bench.py $flags --steps 3 --warmup 1 2> $O/$name.err | tail -1 > $O/bench_$name.json; cat $O/bench_$name.json | dline $name | tee -a $O/decode.txt
done
