export PYTHONPATH=.
timeout 300 python tools/gpu_gemm_check_variant.py 11 2>&1 | tee gpurun_out/check_v11.txt
timeout 600 python tools/gpu_gemm_cold_probe.py 4,8,10,11 2>&1 | tee gpurun_out/cold_probe2.txt
