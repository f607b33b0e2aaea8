export PYTHONPATH=.
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-table gpurun_out/tab_new.txt 2>&1 | tail -1 | cut -c1-330
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-override 2528x4096x28672=10,2528x4096x14336=10,2528x4096x4096=10,2528x4096x6144=10,12000x1024x4096=3,12000x3072x1024=3,12000x1024x1024=3,2528x4096x128256=10,12000x4096x1024=11 --gemm-table gpurun_out/tab_old.txt 2>&1 | tail -1 | cut -c1-330
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-table gpurun_out/tab_new2.txt 2>&1 | tail -1 | cut -c1-330
