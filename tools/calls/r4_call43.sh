# round 4, call 43: C3 again with round 3's step counts (10 timed, 3 warm-up), with and without the per-GEMM HIP events - call 42's 5-step line
# was 17 ms above r03_bench_c3.json while its GEMM time was the same.
export PYTHONPATH=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c43
timeout 80 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4c43/bench_c3.json 2> gpurun_out/r4c43/bench_c3.err
timeout 80 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r4c43/bench_c3_noprof.json 2> gpurun_out/r4c43/bench_c3_noprof.err
python - <<'PY'
import json
for f in ("bench_c3", "bench_c3_noprof"):
    try:
        d = json.loads(open(f"gpurun_out/r4c43/{f}.json").read().strip().split("\n")[-1])
        print(f, round(d["ms_per_step"], 2), (d.get("roofline") or {}).get("gemm_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
