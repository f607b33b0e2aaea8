export PYTHONPATH=.
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for v in 11 18 16 0 4; do timeout 120 python tools/gpu_gemm_check_variant.py $v 2>&1 | grep -v "OK$" | tail -3; done
timeout 500 python tools/gpu_gemm_cold_probe.py 11,18,16 2>&1 | tee gpurun_out/cold_probe_ws.txt
timeout 500 python tools/gpu_gemm_cold_probe.py 11,16,3 enc 2>&1 | tee -a gpurun_out/cold_probe_ws.txt
