export PYTHONPATH=.
for o in "3=1" "3=0" "3=1" "3=0"; do
  echo "== opt $o"; python bench.py --steps 4 --warmup 2 --no-cpu-baseline --opt $o --gemm-table gpurun_out/tab_sup_$o.txt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
