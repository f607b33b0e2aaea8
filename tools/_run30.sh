export PYTHONPATH=.
timeout 300 python tools/gpu_gemm_check_variant.py 12 2>&1 | tee gpurun_out/check_v12.txt
timeout 600 python tools/gpu_gemm_cold_probe.py 8,12 2>&1 | tee gpurun_out/cold_probe3.txt
