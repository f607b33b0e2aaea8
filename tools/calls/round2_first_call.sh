#!/bin/bash
# The probes queued at the end of round 1 (written after that round's GPU budget was spent), in ONE gpurun call so the box
# acquisition is paid once:   gpurun --timeout 900 -- 'bash tools/round2_first_call.sh'
# Each step has its own timeout and log under gpurun_out/round2/; nothing here changes the product path.
R=$PWD; OUT=$R/gpurun_out/round2; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
run tests_gpu       400 python -m pytest tests -m gpu -q -x --timeout 380 -p no:cacheprovider            # baseline: still green?
TAIL=14 run gemm_timeline   120 python tools/gpu_gemm_timeline.py                                         # where the K-tile cycles go
run gemm_timeline_k 120 python tools/gpu_gemm_timeline.py 2528 4096 4096                                 # a single-round shape
run lcp_check       200 python tools/gpu_lcp_check.py                                                    # partial KV-cache reuse on the device
run c3_width        300 python tools/gpu_c3_width_check.py                                               # whisper-large-v3 width parity
run c4_width        300 python tools/gpu_c4_width_check.py                                               # Llama-3.3-70B width through generate()
TAIL=20 run chat_probe      300 python tools/gpu_chat_probe.py 6 48 32                                     # cached vs re-prefilled turns
run bench           200 python bench.py --steps 5 --warmup 2
du -sh $OUT
