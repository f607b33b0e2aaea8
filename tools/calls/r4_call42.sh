# round 4, call 42: C3's per-rank workload (whisper-large-v3 + Llama-3-8B, B = 8) on the round's final build, for the record next to r03_bench_c3.json.
export PYTHONPATH=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c42
timeout 170 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4c42/bench_c3.json 2> gpurun_out/r4c42/bench_c3.err
tail -1 gpurun_out/r4c42/bench_c3.json | cut -c1-400
