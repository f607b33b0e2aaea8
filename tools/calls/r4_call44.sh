# round 4, call 44: is call 42's slow 5-step / 2-warm-up C3 line reproducible (and does C2 show it)?  --gemm-raw off, per-step wall times via --steps 1 repeats are
# not available, so: the same command twice for C3, once for C2.
export PYTHONPATH=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c44
for tag in c3_a c3_b; do
  timeout 60 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4c44/$tag.json 2> gpurun_out/r4c44/$tag.err
done
timeout 60 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4c44/c2.json 2> gpurun_out/r4c44/c2.err
python - <<'PY'
import json
for f in ("c3_a", "c3_b", "c2"):
    try:
        d = json.loads(open(f"gpurun_out/r4c44/{f}.json").read().strip().split("\n")[-1])
        print(f, round(d["ms_per_step"], 2), (d.get("roofline") or {}).get("gemm_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
