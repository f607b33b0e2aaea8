#!/bin/bash
# round 3, call 23: fused attention backward with staged row stores + up-front prologue loads: tests, timeline, bench A/B vs option 13 = 0
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
PYTHONPATH=. timeout 300 python tools/gpu_attn_timeline.py > gpurun_out/r3_attn_timeline_after.txt 2>&1; cat gpurun_out/r3_attn_timeline_after.txt | grep -v amdgpu.ids
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c23_bench_$i.json 2> gpurun_out/r3c23_bench_$i.err
python -c "
import json; d=json.load(open('gpurun_out/r3c23_bench_$i.json')); print('bench $i ms/step %.2f loss %.5f' % (d['ms_per_step'], d['loss']))"
done
