#!/bin/bash
# Round 4, visit M1: the whole GPU suite at the round's final kernels, then the driver's own bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== suite"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -25
  echo "== bench (python bench.py)"; timeout 900 python bench.py 2> gpurun_out/r4m_bench.err | tee gpurun_out/r4m_bench.json | cut -c1-4000
  tail -5 gpurun_out/r4m_bench.err
} 2>&1 | tee gpurun_out/r4_m1.log
