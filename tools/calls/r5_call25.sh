#!/bin/bash
# round 5, call 25: PMC passes (SQ / LDS blocks, separate runs, --kernel-trace only) on the production GEMM of the final build: MFMA-pipe busy fraction,
# wave time split, LDS conflicts at the C2 gate|up shape (variant 31, 256 x 256 tiles)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
VARS="31" PYVARS="31," bash tools/pmc_gemm_prod.sh
cat gpurun_out/pmc_prod/summary.txt
