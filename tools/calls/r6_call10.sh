#!/bin/bash
# round 6, call 10: rope + cache append inside the grouped decode-attention kernel (option 23): tests, decode lines with the separate launch (23=1) vs fused (default)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c10; mkdir -p $O
timeout 900 python -m pytest tests/test_generate_gpu.py tests/test_c4_full_depth_gpu.py tests/test_baseline_configs_gpu.py -q 2>&1 | tail -5 | tee $O/pytest.txt
dline() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', 'decode ms/token', round(r['decode_ms_per_token'],3), 'frac_hbm', round(r['roofline']['frac'],4), 'prefill ms', round(r['prefill_ms'],2), 'tok/s', round(r['value'],1))"; }
for rep in 1 2; do
for f in "c4s_b1_separate:--workload c4s --batch 1 --opt 23=1" "c4s_b1_fused:--workload c4s --batch 1" "c4s_b8_separate:--workload c4s --batch 8 --opt 23=1" "c4s_b8_fused:--workload c4s --batch 8"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 600 python bench.py $flags --steps 3 --warmup 1 2>/dev/null | tail -1 | dline $name | tee -a $O/decode.txt
done
done
for f in "c4_b1_separate:--workload c4 --batch 1 --opt 23=1" "c4_b1_fused:--workload c4 --batch 1" "c4_b8_separate:--workload c4 --batch 8 --opt 23=1" "c4_b8_fused:--workload c4 --batch 8"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 900 python bench.py $flags --steps 3 --warmup 1 2> $O/$name.err | tail -1 > $O/bench_$name.json; cat $O/bench_$name.json | dline $name | tee -a $O/decode.txt
done
