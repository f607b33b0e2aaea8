export PYTHONPATH=.
for v in 18 19 15; do timeout 120 python tools/gpu_gemm_check_variant.py $v 2>&1 | grep -v "OK$" | tail -3; done
timeout 500 python tools/gpu_gemm_cold_probe.py 15,18,16,17,19 2>&1 | tee gpurun_out/cold_probe_ns3.txt
