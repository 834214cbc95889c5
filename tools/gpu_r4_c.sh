#!/bin/bash
# Round 4, visit C: the whole GPU suite on the caller-owned stream workspaces + streamed staging, then the bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== suite"; timeout 1000 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -25
  echo "== bench"; timeout 600 python bench.py --steps 16 --warmup 1 --no-cpu-baseline 2> gpurun_out/r4c_bench.err | tee gpurun_out/r4c_bench.json | cut -c1-3000
  tail -5 gpurun_out/r4c_bench.err
} 2>&1 | tee gpurun_out/r4_c.log
