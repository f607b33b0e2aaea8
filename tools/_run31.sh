export PYTHONPATH=.
timeout 120 python tools/gpu_gemm_check_variant.py 11 2>&1 | tail -11
timeout 120 python tools/gpu_gemm_check_variant.py 12 2>&1 | tail -11
timeout 500 python tools/gpu_gemm_cold_probe.py 8,11,12,13,14 2>&1 | tee gpurun_out/cold_probe_modes.txt
