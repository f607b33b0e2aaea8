export PYTHONPATH=.
for v in 18 19 20; do timeout 120 python tools/gpu_gemm_check_variant.py $v 2>&1 | grep -v "OK$" | tail -3; done
timeout 500 python tools/gpu_gemm_cold_probe.py 11,18,15,19,16,20 2>&1 | tee gpurun_out/cold_probe_bal.txt
