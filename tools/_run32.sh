export PYTHONPATH=.
for v in 15 16 17 11; do timeout 120 python tools/gpu_gemm_check_variant.py $v 2>&1 | grep -v "OK$" | tail -3; done
timeout 500 python tools/gpu_gemm_cold_probe.py 10,11,15,16 2>&1 | tee gpurun_out/cold_probe_ph8.txt
timeout 500 python tools/gpu_gemm_cold_probe.py 3,8,11,15,16,17 enc 2>&1 | tee gpurun_out/cold_probe_ph8_enc.txt
