export PYTHONPATH=.
for v in 23 24 25 26 11 18; do timeout 150 python tools/gpu_gemm_check_variant.py $v 2>&1 | grep -v "OK$" | tail -3; done
timeout 300 python tools/gpu_gemm_overhead_probe.py 11,23 2>&1 | grep -v amdgpu
timeout 400 python tools/gpu_gemm_cold_probe.py 11,23,16,24 2>&1 | grep -v amdgpu
timeout 400 python tools/gpu_gemm_cold_probe.py 11,23,16,24 enc 2>&1 | grep -v amdgpu
