#!/bin/bash
# Round 2, call 24: last sanity pass on the final build (ABI 8): full -m gpu suite, smoke, default bench line.
R=$PWD; OUT=$R/gpurun_out/r2c24; mkdir -p $OUT; export PYTHONPATH=$R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 400 python bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-330
