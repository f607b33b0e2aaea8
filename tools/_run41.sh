export PYTHONPATH=.
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for v in 11 18 16 17 0 4 8 10; do timeout 120 python tools/gpu_gemm_check_variant.py $v 2>&1 | grep -v "OK$" | tail -3; done
timeout 300 python tools/gpu_gemm_overhead_probe.py 11 2>&1 | grep -v amdgpu
for o in "1=2" "1=1" "1=2" "1=1"; do
  echo "== opt $o"; python bench.py --steps 4 --warmup 2 --no-cpu-baseline --opt $o 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
