#!/bin/bash
# round 6, call 7: the spill-free build (act 2 / 3 epilogues as their own kernel instantiations, LoRA epilogue removed) + the NN dgrad form: tests, the NN-vs-NT
# cold probe, step lines (CE with resident W^T / streamed / NN; the recipe flavours old vs new)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c7; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_lora_gpu.py tests/test_generate_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -6 | tee $O/pytest.txt
timeout 900 python tools/gpu_gemm_nn_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/nn_probe.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4), 'gemm_ms', round((r.get('roofline') or {}).get('gemm_ms_per_step', 0),2), 'GiB', round(r.get('resident_gib', 0), 1))"; }
for rep in 1 2; do
for f in "ce:" "ce_nn:--dgrad-nn" "ce_stream:--stream-wt on" "lora8_old:--audio-lora-r 8 --opt 19=1,20=1,21=1" "lora8_new:--audio-lora-r 8" "kl_lora8_old:--loss kl --audio-lora-r 8 --opt 19=1,20=1,21=1" "kl_lora8_new:--loss kl --audio-lora-r 8"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>/dev/null | tail -1 | line $name | tee -a $O/flavours.txt
done
done
