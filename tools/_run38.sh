export PYTHONPATH=.
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_f32_parity_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for v in 11 18 16 0 4 8; do timeout 120 python tools/gpu_gemm_check_variant.py $v 2>&1 | grep -v "OK$" | tail -3; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm-table gpurun_out/tab_ws.txt 2>&1 | tail -1 | cut -c1-260
