#!/bin/bash
# Round 3, call A: stream-K GEMM parity + sweep, HBM calibration kernels, knob-17 default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== pytest -m gpu: gemm + relpos kernels"
  timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "gemm or relpos" 2>&1 | tail -8
  echo "== stream calibration"
  timeout 200 python tools/microbench.py --stream 2>&1 | tail -8
  echo "== stream-K sweep"
  timeout 600 python tools/microbench.py --sk 2>&1 | tail -170
  echo "== relpos-t"
  timeout 200 python tools/microbench.py --relpos-t 2>&1 | tail -4
} 2>&1 | tee gpurun_out/r3_a.log
