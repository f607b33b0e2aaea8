export PYTHONPATH=.
for o in "1=1,2=1" "1=0,2=1" "1=1,2=0" "1=0,2=0" "1=1,2=1"; do
  echo "== opt $o"; python bench.py --steps 4 --warmup 2 --no-cpu-baseline --opt $o --gemm-table gpurun_out/tab_opt_$o.txt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
