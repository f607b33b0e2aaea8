#!/bin/bash
# round 3, call 37: streamed weight transposes with non-temporal accesses (default) vs plain ones (option 5 = 1) vs resident copies, same box
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "streamed" 2>&1 | grep -E "passed|failed|rror" | tail -3
run() { timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 > gpurun_out/r3c37_$1.json 2> gpurun_out/r3c37_$1.err
  python -c "
import json; d=json.load(open('gpurun_out/r3c37_$1.json')); print('%-14s ms/step %.2f gemm frac %.4f loss %.5f' % ('$1', d['ms_per_step'], d['roofline']['frac'], d['loss']))" || tail -3 gpurun_out/r3c37_$1.err; }
run warm "--stream-wt off"
run resident "--stream-wt off"
run stream_nt "--stream-wt on"
run stream_plain "--stream-wt on --opt 5=1"
run resident_b "--stream-wt off"
run stream_nt_b "--stream-wt on"
run stream_plain_b "--stream-wt on --opt 5=1"
