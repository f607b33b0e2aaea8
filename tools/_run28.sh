export PYTHONPATH=.
python -m pytest tests/test_kl_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -5
python tools/gpu_gemm_cold_probe.py 2>&1 | tee gpurun_out/cold_probe.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-table gpurun_out/tab_base.txt 2>&1 | tail -1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-override 2528x4096x14336=10,2528x4096x4096=10,2528x4096x6144=10 --gemm-table gpurun_out/tab_v10.txt 2>&1 | tail -1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-override 2528x4096x14336=6,2528x4096x4096=6,2528x4096x6144=6,2528x4096x28672=6 --gemm-table gpurun_out/tab_v6.txt 2>&1 | tail -1
